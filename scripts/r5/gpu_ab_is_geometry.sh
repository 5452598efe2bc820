cd /root/repo
for round in 1 2; do for v in base is14 is24; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
  for rays in 512 1024; do python bench.py --mode render --rays $rays --steps 100 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $round, 'rays $rays', 'ms/step %.4f' % d['ms_per_step'])"; done
done; done
