cd $GRAFT_REPO_ROOT
for o in "" "--config 2m"; do
  echo "== $o"; python3 bench.py --steps 20 --warmup 3 --no-extras $o 2>&1 | grep -E "^\{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l);print(round(d['ms_per_step'],4),{k:round(v,4) for k,v in d['phases_ms'].items()},d['kept_rank0'])"
done
