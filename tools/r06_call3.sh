#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c3; mkdir -p $O
MFX_LIB_PATH=build_variants/lib_probe.so timeout 300 python tools/probes/dcn_lds_probe.py module > $O/probe_module.txt 2>&1
MFX_LIB_PATH=build_variants/lib_probe.so timeout 300 python tools/probes/dcn_lds_probe.py kernel > $O/probe_kernel.txt 2>&1
MFX_LIB_PATH=build_variants/lib_probe.so timeout 300 python tools/probes/dcn_lds_probe.py module 2 96 320 64 > $O/probe_module_b2.txt 2>&1
