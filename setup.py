from setuptools import find_packages, setup

setup(
    name="locator-amd",
    version="0.1.0",
    description="MI355X-native implementation of kr-colab/locator's training path",
    packages=find_packages(include=["locator_amd", "locator_amd.*"]),
    package_data={"locator_amd": ["liblocator_hip.so", "csrc/*"]},
    entry_points={"console_scripts": ["locator=locator_amd.locator:main"]},
    python_requires=">=3.8",
)
