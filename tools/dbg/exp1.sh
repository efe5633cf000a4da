cd $GRAFT_REPO_ROOT
for o in "" "--opt seg_unite=0" "--opt seg_unite=0 --opt seg_dbg=1" "--opt seg_blocks=4096" "--opt seg_blocks=6144"; do
  echo "== $o"; UMIHIP_TIMING=1 python3 bench.py --steps 20 --warmup 3 --no-extras $o 2>&1 | grep -E "^seg_pair|^\{" | python3 -c "
import json,sys
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l);print(round(d['ms_per_step'],4),{k:round(v,4) for k,v in d['phases_ms'].items()},d['kept_rank0'])
  else: print(l.strip())"
done
