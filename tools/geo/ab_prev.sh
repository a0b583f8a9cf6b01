# same-box A/B of the built library against tools/geo/variants/prev.so on the hinted and the cold headline frame
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for V in prev new; do
  if [ $V = prev ]; then export ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/geo/variants/prev.so; else unset ENVIDR_AMD_LIB; fi
  for MODE in "" "--cold"; do
  timeout 300 python bench.py --headline-only $MODE --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$V', '$MODE' or 'hinted', 'ms/frame %.3f geometry %.3f shading %.3f' % (j['ms_per_step'], j['frame']['geometry_ms'], j['frame']['shading_ms']))"
  done
done
done
