O=gpurun_out/r4_ctl; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or sampled or staged or sharded_local_c_abi or controller or sampling or zero_score or extrapolated or larger_random" > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
timeout 600 python tools/ctl_probe.py --staged --sets "overlap=1;overlap=1,period=64;period=64;period=128;overlap=1,period=128" > $O/ctl_default.log 2>&1; cat $O/ctl_default.log
DHR_ADAPTIVE_RANK=0 timeout 400 python tools/ctl_probe.py --staged --sets "overlap=1" > $O/ctl_noadapt.log 2>&1; cat $O/ctl_noadapt.log
DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_nont.so timeout 400 python tools/ctl_probe.py --sets "overlap=1" > $O/ctl_nont.log 2>&1; cat $O/ctl_nont.log
