mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or fuse or group or bottleneck" > gpurun_out/conv_tests.log 2>&1 || { tail -20 gpurun_out/conv_tests.log; exit 1; }
tail -1 gpurun_out/conv_tests.log
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
python tools/sq_summary.py r04j_tmp > /dev/null && mv profiles/r04j_tmp_sq_summary.json gpurun_out/r04j_sq_summary.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j_sq_summary.json'))
for k,v in d.items():
    if isinstance(v,dict):
        for kk,vv in v.items():
            if isinstance(vv,dict) and 'lds_bank_conflict_frac' in vv: print(kk[:50], vv['lds_bank_conflict_frac'], vv.get('mfma_busy_frac'))
PY
