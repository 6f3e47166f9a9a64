#!/bin/bash
# BOUNDARY PROOF (build container only): the reference's own test, /root/reference/samples/tests/test_lineartet.cpp, compiled IN PLACE and
# UNCHANGED against this repository's mirror of the reference API (admm-elastic_amd/host/include: Solver.hpp, TetEnergyTerm.hpp ...) with the
# mirror's value types switched to the Eigen the reference vendors (-DADMM_WITH_EIGEN, found at build time under /root/reference/deps/Eigen3,
# never copied) and linked with libadmm_hip.so.  Every update() / step() of the test runs in the HIP kernels.
#   tests/cpp/compat/MCL/{Vec,XForm}.hpp: the two mclscene headers the test includes (mclscene is an empty submodule in /root/reference)
#   output: oracle/_ref/test_lineartet_reference (git-ignored like every artefact made from reference sources; it travels to the GPU box,
#   where tests/test_cpp_api.py::test_reference_lineartet_unchanged runs it and expects "SUCCESS")
# Compiling and linking need no GPU; running does.
set -e
cd "$(dirname "$0")/../.."
REF=/root/reference
[ -f $REF/samples/tests/test_lineartet.cpp ] || { echo "no reference tree: nothing to build"; exit 0; }
python -c "import sys; sys.path.insert(0, 'admm-elastic_amd'); import build; build.build_library()" > /dev/null
mkdir -p oracle/_ref
g++ -std=c++17 -O2 -w -DADMM_WITH_EIGEN -I$REF/deps/Eigen3 -Itests/cpp/compat -Iadmm-elastic_amd/host/include -Iinclude \
    $REF/samples/tests/test_lineartet.cpp admm-elastic_amd/host/src/Solver.cpp \
    -Ladmm-elastic_amd -ladmm_hip -Wl,-rpath,'$ORIGIN/../../admm-elastic_amd' -o oracle/_ref/test_lineartet_reference
echo "built oracle/_ref/test_lineartet_reference (reference test, unchanged, against the mirror)"
