cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/attn_one.py <<PY
import sys, os
sys.path.insert(0, "$R"); sys.argv=["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("ab", "$R/tools/attn_bench.py"); ab = importlib.util.module_from_spec(spec); spec.loader.exec_module(ab)
ab.run(144, 1024, 1, reps=3)
PY
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc1 -- python /tmp/attn_one.py > $R/gpurun_out/attn_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc2 -- python /tmp/attn_one.py > $R/gpurun_out/attn_pmc2.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("attn_pmc1","attn_pmc2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        for k,v in agg.items():
            if "attn_mfma" in k:
                print(d, k, {c: round(x/ (4*1),1) for c,x in v.items()})
PY
tail -3 gpurun_out/attn_pmc2.log
