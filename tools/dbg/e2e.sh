cd $GRAFT_REPO_ROOT
python3 tools/make_bam.py /tmp/e2e_in.bam --reads 2000000 --positions 20000 > /dev/null 2>&1
TIMEFORMAT="wall %R s"
for w in gpu host; do for r in 1 2; do
{ time ./umi_collapse_rs_amd/bin/umicollapse -i /tmp/e2e_in.bam -o /tmp/e2e_out_$w.bam --merge avgqual --num-threads 16 --stage $w ; } 2>&1 | grep -E "phases|finished in|wall" | tr '\n' ' '; echo; done; done
cmp /tmp/e2e_out_gpu.bam /tmp/e2e_out_host.bam && echo same
