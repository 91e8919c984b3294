R=$PWD; OUT=gpurun_out/sq_words; mkdir -p $R/$OUT; cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$OUT/pass$i -- python $R/tools/sweep.py --steps 4 --warmup 1 --inputs 4 --configs strict:16:1 > $R/$OUT/pass$i.log 2>&1
done
cd $R
for k in k_sync_words k_sync_slots k_gather_rows_call k_sync_orbit_global; do python tools/summarize_sq.py $k $(ls $OUT/pass*/*/*counter_collection.csv) > $OUT/$k.json; done
rm -rf $OUT/pass*/
python - <<'PY'
import json
for k in ("k_sync_words","k_sync_slots","k_gather_rows_call","k_sync_orbit_global"):
    d=json.load(open(f"gpurun_out/sq_words/{k}.json"))
    print(k, {x:d["per_launch"].get(x) for x in ("SQ_WAVES","SQ_INSTS_VALU","SQ_ACTIVE_INST_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES")}, d["derived"])
PY
