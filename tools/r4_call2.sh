#!/bin/bash
# round 4 call 2: (a) where does the precision-5 Kokoro test fault (blocking launches), (b) does staggering the two resident workgroups of a CU pay (MI355_CONV_WS_STAGGER sweep)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_kokoro_gpu.py -q -x -s -k "precision5" > $O/pytest_kokoro_p5_blocking.txt 2>&1; echo "kokoro_p5 rc=$?" > $O/rc.txt
for n in 0 3 6 10; do
  MI355_CONV_WS_STAGGER=$n timeout 300 python tools/bench_conv.py --prec-ab --batch 32 --rounds 5 --out $O/conv_prec_ab_stagger$n.txt > /dev/null 2> $O/conv_stagger$n.err; echo "stagger $n rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; grep -v "^  File\|^Extension" $O/pytest_kokoro_p5_blocking.txt | head -30
python - <<'PY'
import os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
rows={}
for n in (0,3,6,10):
    for ln in open(O+"/conv_prec_ab_stagger%d.txt"%n).read().splitlines()[1:]:
        f=ln.split(); key=tuple(f[:4])+(f[6],); rows.setdefault(key,{})[n]=f[7]
print("cin cout k dil variant | ms at stagger 0 / 3 / 6 / 10")
for k,v in rows.items(): print(" ".join(k), "|", " ".join(v.get(n,"-") for n in (0,3,6,10)))
PY
