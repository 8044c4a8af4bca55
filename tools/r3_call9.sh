#!/bin/bash
# round 3 call 9: the decode configs behind bench.py (--config qwen3 | csm | whisper) through ShardChannel at world 1 (same value as the tools), the
# one-rank RCCL call test with the new int64 / dense collectives, full GPU suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -m pytest tests/test_shard_nccl_gpu.py -q -m gpu > $O/t_nccl.log 2>&1; echo "nccl rc=$?" > $O/rc.txt
for c in qwen3 csm whisper; do
  timeout 900 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?" >> $O/rc.txt
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --config csm --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_csm_tr1.json 2> $O/bench_csm_tr1.err; echo "csm torchrun1 rc=$?" >> $O/rc.txt
export MI355_MARGIN_REPORT=$O/margin_report.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/t_full.log 2>&1; echo "full rc=$?" >> $O/rc.txt
timeout 900 python bench.py --no-pmc > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
tail -3 $O/t_nccl.log; cat $O/rc.txt; tail -12 $O/t_full.log
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for c in ("qwen3","csm","whisper","csm_tr1"):
    try:
        d=json.load(open(O+"/bench_%s.json"%c)); print(c, round(d["value"],1), d["unit"], "n_gpus", d["n_gpus"], d.get("ms_per_frame"), d["config"].get("parallelism"))
    except Exception as e: print(c, "ERR", e, open(O+"/bench_%s.err"%c).read()[-400:])
d=json.load(open(O+"/bench_default.json")); print("kokoro", round(d["value"]/1e6,1), d["ms_per_step"], d["roofline"]["frac"], d.get("latency_b1"))
PY
