lscpu | grep -i "numa\|socket\|model name" | head -12
for c in /sys/class/drm/card*/device/numa_node; do echo $c $(cat $c); done 2>/dev/null | head
python - <<'PY'
import os
print("affinity of this process:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], "...")
PY
