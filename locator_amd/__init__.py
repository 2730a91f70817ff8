"""locator_amd — MI355X-native implementation of kr-colab/locator's genotype -> coordinate
regression training path.  Host code is Python; all device arithmetic is hand-written HIP for
gfx950 behind the C ABI declared in include/locator_hip.h (liblocator_hip.so).

There is deliberately no CPU fallback: importing the device layer without the built
extension, or constructing a network without a GPU, raises.
"""
__version__ = "0.1.0"
