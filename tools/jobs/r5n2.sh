cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5n2
WISKI_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r5n2/out.txt 2> gpurun_out/r5n2/err.txt
echo rc=$?; tail -1 gpurun_out/r5n2/out.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['n_gpus'], r['ms_per_step'], r['config']['parallelism'][:120]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in r['extra'].items() if 'exchange' in k or 'error' in k or 'sharded' in k or 'alone' in k})"
tail -3 gpurun_out/r5n2/err.txt
