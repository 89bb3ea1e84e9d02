# PMC pass over lk_kernel (tools/lk_exp.py): instruction mix and wait cycles per wave
export TMPDIR=/tmp PLV_STREAM_CACHE=/tmp/lk_stream.npz
REPO=$(pwd); OUT=gpurun_out/lkpmc; mkdir -p $OUT
python3 tools/lk_exp.py C 0 > $OUT/plain.log 2>&1
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $REPO/$OUT/$tag -o run -- python3 $REPO/tools/lk_exp.py C 0 > $REPO/$OUT/$tag.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/lkpmc/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "lk_kernel" in r["Kernel_Name"] and "lk_kernel<" in r["Kernel_Name"] or "lk_wave" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v)//4:]
        print(f"{k:28s} per dispatch {sum(v)/len(v):14.0f}   ({len(v)} dispatches)")
PY
rm -rf $OUT/SQ_*/
