set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcw; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -o "SQ_[A-Z_0-9]*" $OUT/avail.txt | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
QP="python3 $R/tools/quick_perf.py 6144"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- $QP > $OUT/p1.out 2> $OUT/p1.err || tail -5 $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -- $QP > $OUT/p2.out 2> $OUT/p2.err || tail -5 $OUT/p2.err
python3 $R/tools/pmc_sq_summary.py $OUT/p1 $OUT/p2 > $OUT/wait.json || true
cat $OUT/wait.json
rm -rf $OUT/p1 $OUT/p2
