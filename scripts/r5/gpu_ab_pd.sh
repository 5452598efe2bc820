cd /root/repo
for round in 1 2 3; do for v in base ${VARIANTS:-pd3}; do
  if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
  python bench.py --mode render --rays 512 --steps 100 --warmup 10 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', $round, 'ms/step %.4f' % d['ms_per_step'])"
done; done
