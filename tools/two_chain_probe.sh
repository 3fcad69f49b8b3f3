# would two independent half-batch chains fill the chip better than one?  one process at bs 256, one at bs 128, two at bs 128 side by side
cd /root/repo; mkdir -p gpurun_out
O=gpurun_out/two_chain; mkdir -p $O
B="python bench.py --steps 60 --warmup 15 --no-extra --no-cpu-baseline"
$B --batch 256 > $O/one_256.json 2> $O/one_256.err
$B --batch 128 > $O/one_128.json 2> $O/one_128.err
$B --batch 128 > $O/two_a.json 2> $O/two_a.err &
P1=$!
$B --batch 128 > $O/two_b.json 2> $O/two_b.err &
P2=$!
wait $P1 $P2
$B --steps 120 --batch 128 > $O/two_c.json 2> $O/two_c.err &
P1=$!
$B --steps 120 --batch 128 > $O/two_d.json 2> $O/two_d.err &
P2=$!
wait $P1 $P2
python - <<PY
import json
for n in ("one_256","one_128","two_a","two_b","two_c","two_d"):
    try:
        d=json.load(open("$O/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d["ms_per_step_p50"])
    except Exception as e: print(n, "FAILED", e)
PY
