cd $GRAFT_REPO_ROOT
bash tools/quick_prof.sh 2>&1 | grep -E "seg_sc|seg_count|prep_kernel|^0\."
