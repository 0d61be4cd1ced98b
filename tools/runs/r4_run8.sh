O=gpurun_out/r4_t8; mkdir -p $O
timeout 600 python tools/gather_probe.py dense 2>&1 | grep -v amdgpu.ids | tee $O/gather_dense.log
timeout 600 python tools/gather_probe.py hybrid 2>&1 | grep -v amdgpu.ids | tee $O/gather_hybrid.log
