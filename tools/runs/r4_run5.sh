O=gpurun_out/r4_t5; mkdir -p $O
timeout 600 python tools/shard_sim.py 2>&1 | grep -v amdgpu.ids | tee $O/shard_sim_default.log
DHR_ADAPTIVE_RANK=0 timeout 600 python tools/shard_sim.py 2>&1 | grep -v amdgpu.ids | tee $O/shard_sim_noadapt.log
timeout 600 python tools/shard_sim.py --main-chunks 3 2>&1 | grep -v amdgpu.ids | tee $O/shard_sim_c3.log
