#!/bin/bash
# What the GPU box exposes about clocks / power / partition state (for the "slow mode" question).
echo "== rocm-smi"; rocm-smi --showclocks --showpower --showperflevel --showtemp --showmemuse 2>&1 | head -60
echo "== rocm-smi partitions"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20
echo "== amd-smi"; (amd-smi metric -g 0 --clock --power --temperature --perf-level 2>&1 || true) | head -80
echo "== sysfs"
for d in /sys/class/drm/card*/device; do
  [ -f $d/pp_dpm_sclk ] || continue
  echo "-- $d"; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk power_dpm_force_performance_level current_compute_partition current_memory_partition gpu_busy_percent; do
    [ -r $d/$f ] && { echo "[$f]"; cat $d/$f; }; done
  for h in $d/hwmon/hwmon*; do for f in power1_average power1_input power1_cap temp1_input temp2_input freq1_input freq2_input; do [ -r $h/$f ] && echo "$f=$(cat $h/$f)"; done; done
done
echo "== rocminfo (agent names, clocks)"; rocminfo 2>&1 | grep -iE "Marketing|Max Clock|Compute Unit|Name:.*gfx|Partition" | head -20
