O=gpurun_out/r4_t12; mkdir -p $O
for i in 1 2; do
timeout 400 python tools/ctl_probe.py --reps 6 > $O/ctl_h64_$i.log 2>&1; grep defaults $O/ctl_h64_$i.log | sed 's/^/h64 /'
DHR_HIP_LIB=$PWD/dhr_amd/csrc/_ab/libdhr_hip_h32.so timeout 400 python tools/ctl_probe.py --reps 6 > $O/ctl_h32_$i.log 2>&1; grep defaults $O/ctl_h32_$i.log | sed 's/^/h32 /'
done
