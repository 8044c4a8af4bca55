#!/bin/bash
# round 3 call 10: EnCodec decode (lstm_seq + conv schedule) parity and its bench line; the two tests fixed after call 9
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_shard_nccl_gpu.py tests/test_kitten_gpu.py -q -m gpu > $O/t_enc.log 2>&1; echo "enc rc=$?" > $O/rc.txt
timeout 600 python tools/bench_codecs.py --only encodec > $O/bench_encodec.json 2> $O/bench_encodec.err; echo "bench rc=$?" >> $O/rc.txt
tail -25 $O/t_enc.log; cat $O/rc.txt; head -c 900 $O/bench_encodec.json; tail -3 $O/bench_encodec.err
