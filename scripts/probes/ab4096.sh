for round in 1 2; do for v in chunk8k chunk16k; do
 if [ "$v" = base ]; then unset EMAP_HIP_LIB; else export EMAP_HIP_LIB=$PWD/emap_amd/lib/$v/libemap_hip.so; fi
 python bench.py --mode train --rays 4096 --steps 20 --warmup 5 --graph off --no-cpu-baseline --no-other-modes --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', $round, 'ms/step %.3f'%d['ms_per_step'], d.get('backward_kernels'))"
done; done
