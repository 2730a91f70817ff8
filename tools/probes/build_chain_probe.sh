#!/bin/bash
# Measurement builds of the chained layer-1 kernel with the round-6 probe switches (docs/history/round6.md section 1).  The
# product's l1_chain.hip does not carry them (an A / B on one box put the instrumented source 0.6 % behind the plain one in
# samples/s at equal kernel time): tools/probes/l1_chain_probes.patch adds them to a COPY of the source.
#   bash tools/probes/build_chain_probe.sh TAG -DLOC_CHAIN_ALT=1          -> build/liblocator_hip_TAG.so
# switches: LOC_CHAIN_ALT=1 (odd steps walk their k-tiles last first), LOC_CHAIN_STAMPS=<workgroup> (cycle stamps,
# tools/probes/chain_stamps.py), LOC_CHAIN_STAGGER=n, LOC_CHAIN_WSTAG=n (+ LOC_CHAIN_ABLATE=1), LOC_CHAIN_EARLY=1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd $R/locator_amd/csrc || exit 1
make > /dev/null 2>&1
cp l1_chain.hip /tmp/l1_chain_probe_$TAG.hip
( cd /tmp && patch -s -o /tmp/l1_chain_probe_$TAG.patched.hip /tmp/l1_chain_probe_$TAG.hip $R/tools/probes/l1_chain_probes.patch ) || exit 1
cp /tmp/l1_chain_probe_$TAG.patched.hip ./l1_chain_probe_tmp_$TAG.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c l1_chain_probe_tmp_$TAG.hip -o /tmp/l1_chain_probe_$TAG.o
rc=$?
rm -f l1_chain_probe_tmp_$TAG.hip
[ $rc -eq 0 ] || exit $rc
mkdir -p $R/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/liblocator_hip_$TAG.so $(ls *.o | grep -v '^l1_chain.o$') /tmp/l1_chain_probe_$TAG.o
