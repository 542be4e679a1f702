#!/bin/bash
# Re-collects everything under profiles/ that depends on the kernels (run on the GPU box from the repository root):
# counters of the headline launch, the 10 M-triangle terrain at 4K and a path-tracing pass; kernel statistics; the
# launches of one pass in order. Output under gpurun_out/refresh/.
set -e
export TMPDIR=/tmp
O=gpurun_out/refresh; mkdir -p $O
bash tools/pmc_collect.sh headline python3 tools/closest_probe.py --scene soup > $O/pmc_headline.log 2>&1
python3 tools/pmc_json.py headline --skip 1 > $O/pmc_headline_raw.json
echo headline counters done
bash tools/pmc_collect.sh terrain10m python3 tools/closest_probe.py --scene terrain --width 3840 --height 2160 > $O/pmc_terrain.log 2>&1
python3 tools/pmc_json.py terrain10m --skip 1 > $O/pmc_terrain10m_raw.json
echo terrain counters done
# (one pass in flight, so that the kernel trace behind the counter passes times every kernel alone: a switch of the tuning build, same kernels)
RTX_LIBRARY=tuning RTX_PIPELINE=1 bash tools/pmc_collect.sh pathtrace python3 tools/pt_profile.py --single 0 --passes 12 > $O/pmc_pt.log 2>&1
python3 tools/pmc_json.py pathtrace > $O/pmc_pathtrace_raw.json
echo pathtrace counters done
TAG="${PROFILE_TAG:-r03}"
python3 tools/pmc_build.py --headline $O/pmc_headline_raw.json --big $O/pmc_terrain10m_raw.json --pathtrace $O/pmc_pathtrace_raw.json --passes 16 --tag $TAG --out $O/pmc.json > /dev/null
echo counter summary built
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/trace_headline -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --only headline > $O/trace_headline.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace_headline -name '*.db' | head -1) --csv > $O/closest_kernel_stats.csv
echo headline trace done
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/trace_pt -- python3 tools/pt_profile.py --single 0 --passes 32 > $O/trace_pt.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace_pt -name '*.db' | head -1) --csv > $O/pathtrace_kernel_stats.csv
echo pathtrace trace done
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/trace_single -- python3 tools/pt_profile.py --single 4 --passes 0 > $O/trace_single.log 2>&1
python3 tools/pt_depths.py $(find $O/trace_single -name '*.db' | head -1) > $O/pt_launches.txt
echo single-pass trace done
# instant radiosity at 128 light paths per pass (8 x 16, what tools/pt_profile.py sets): counters and kernel statistics of the gather
RTX_LIBRARY=tuning RTX_PIPELINE=1 bash tools/pmc_collect.sh radiosity python3 tools/pt_profile.py --algorithm 2 --single 0 --passes 6 > $O/pmc_ir.log 2>&1
python3 tools/pmc_json.py radiosity > $O/pmc_radiosity_raw.json
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/trace_ir -- python3 tools/pt_profile.py --algorithm 2 --single 0 --passes 12 > $O/trace_ir.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace_ir -name '*.db' | head -1) --csv > $O/radiosity_kernel_stats.csv
echo radiosity done
