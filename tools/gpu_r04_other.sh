#!/bin/bash
# Round 4: the other BASELINE configs on the final build (bench lines + per-op tables), and the rocprofv3 kernel stats of the
# CrowdPose flip test (configs[4]: TRP attention / project / tail / SGM lines) and of HRNet-W48 384x288 (configs[2]).
TAG=${TAG:-r04g}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
one() { n=$1; shift; timeout -k 10 300 python bench.py "$@" > gpurun_out/$n.log 2>&1 || { tail -5 gpurun_out/$n.log; exit 1; }; tail -1 gpurun_out/$n.log > gpurun_out/$n.json; python - gpurun_out/$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d["dtype"], d["config"].get("launches_per_step"), (d.get("roofline", {}).get("whole_step") or {}).get("frac"), (d.get("single_stream") or {}).get("value"))
PY
}
one ${TAG}_bench_bf16_w32 --workload hrnet_w32_256x192 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline || exit 1
one ${TAG}_bench_w48 --workload hrnet_w48_384x288 --steps 20 --warmup 5 --no-cpu-baseline --per-op gpurun_out/${TAG}_per_op_hrnet_w48_384x288.json || exit 1
one ${TAG}_bench_rsg --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline --per-op gpurun_out/${TAG}_per_op_rsgnet_w32_256x192_coco.json || exit 1
one ${TAG}_bench_rsg48 --workload rsgnet_w48_384x288_coco --steps 10 --warmup 3 --no-cpu-baseline || exit 1
one ${TAG}_bench_cp --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline --per-op gpurun_out/${TAG}_per_op_rsgnet_crowdpose_fliptest.json || exit 1
one ${TAG}_bench_train_hrnet --train --steps 10 --warmup 3 --no-cpu-baseline || exit 1
one ${TAG}_bench_train_rsgnet --train --workload rsgnet_w32_256x192_coco --steps 10 --warmup 3 --no-cpu-baseline || exit 1
rm -rf gpurun_out/prof_cp gpurun_out/prof_w48
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cp -o trace -- python3 bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_cp.log 2>&1 || { tail -5 gpurun_out/prof_cp.log; exit 1; }
find gpurun_out/prof_cp -name '*kernel_trace.csv' -delete
cp $(find gpurun_out/prof_cp -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_rsgnet_crowdpose_fliptest_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w48 -o trace -- python3 bench.py --workload hrnet_w48_384x288 --steps 10 --warmup 3 --streams 1 --no-cpu-baseline > gpurun_out/prof_w48.log 2>&1 || { tail -5 gpurun_out/prof_w48.log; exit 1; }
find gpurun_out/prof_w48 -name '*kernel_trace.csv' -delete
cp $(find gpurun_out/prof_w48 -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_hrnet_w48_384x288_kernel_stats_one_stream.csv
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_sq_cp.log 2>&1 || { tail -5 gpurun_out/pmc_sq_cp.log; exit 1; }
python tools/sq_summary.py ${TAG}_tmp > /dev/null && mv profiles/${TAG}_tmp_sq_summary.json gpurun_out/${TAG}_sq_summary_rsgnet_crowdpose_fliptest.json
head -8 gpurun_out/${TAG}_rsgnet_crowdpose_fliptest_kernel_stats.csv | cut -c1-150
echo visit done
# the raw counter / trace dumps stay on the box (gpurun copies back at most 64 MiB): the summaries above are what is kept
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/prof gpurun_out/prof_cp gpurun_out/prof_w48
