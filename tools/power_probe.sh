#!/bin/bash
# samples the GPU's power / clock sysfs files while a bench runs (read-only)
O=gpurun_out; mkdir -p $O
ls /sys/class/drm/ > $O/power_probe.txt 2>&1
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H" >> $O/power_probe.txt
ls $H >> $O/power_probe.txt 2>&1
D=$(dirname $(dirname $H))
( timeout -k 10 200 python bench.py --steps ${1:-200} --warmup 2 --no-cpu-baseline > $O/bench_power.json 2> $O/bench_power.err ) &
BP=$!
sleep 25
for i in $(seq 1 40); do
  echo "$(date +%s.%N) p_avg=$(cat $H/power1_average 2>/dev/null) p_in=$(cat $H/power1_input 2>/dev/null) cap=$(cat $H/power1_cap 2>/dev/null) sclk=$(cat $H/freq1_input 2>/dev/null) temp=$(cat $H/temp1_input 2>/dev/null) $(grep '\*' $D/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')" >> $O/power_probe.txt
  sleep 0.2
done
wait $BP
python3 -c "import json; d=json.loads(open('$O/bench_power.json').read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
