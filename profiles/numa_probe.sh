#!/bin/bash
# Where is the GPU, where do the CPUs sit, and what does the export's link rate look like from either side?  (one gpurun call)
export TMPDIR=/tmp
for c in /sys/class/drm/card*/device/numa_node; do echo "$c: $(cat $c)"; done
for n in /sys/devices/system/node/node*; do echo "$n cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4 $5}')"; done
nproc; grep Cpus_allowed_list /proc/self/status
python3 - <<'PY'
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int(-1)
for name, attr in (("HostNumaId", None),):
    pass
PY
python3 profiles/measure_cli.py 10000 > /dev/null 2>&1   # writes nothing new: the FASTA exists or is made at full size below
python3 profiles/measure_cli.py > /tmp/base.txt 2>&1
for n in /sys/devices/system/node/node*; do
  cpus=$(cat $n/cpulist)
  echo "== taskset -c $cpus"
  taskset -c $cpus python3 profiles/measure_cli.py 2>&1 | grep -o "stream to the host and into the slots [0-9.]*\|staging [0-9.]*\|build_debruijn_graph [0-9.]*\|read+parse [0-9.]*"
done
