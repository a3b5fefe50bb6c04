#!/bin/bash
# what can an ordinary user read about socket power / sclk on the GPU box?  (input for scripts/power_trace.py)
echo "== hwmon"; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $d; ls $d 2>/dev/null | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input; do [ -r $d/$f ] && echo "$f=$(cat $d/$f 2>&1)"; done; done
echo "== pp_dpm"; for c in /sys/class/drm/card*/device; do [ -r $c/pp_dpm_sclk ] && { echo $c; cat $c/pp_dpm_sclk; }; [ -r $c/gpu_busy_percent ] && echo "busy=$(cat $c/gpu_busy_percent)"; done 2>&1 | head -40
echo "== rocm-smi"; which rocm-smi amd-smi; (time rocm-smi --showpower --showclocks --json) 2>&1 | head -40
echo "== amd-smi"; (time amd-smi metric -p -c --json) 2>&1 | head -80
