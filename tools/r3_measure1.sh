#!/bin/bash
# round 3, GPU call 2: the rest of the GPU suite, the headline bench (with the secondary configs), plain vs block recomputation
cd "$(dirname "$0")/.."
o=gpurun_out/r3m1; mkdir -p $o
python -m pytest tests/test_dit_gpu.py tests/test_fp8_gpu.py tests/test_gemm_gpu.py tests/test_gradsync_overlap_gpu.py tests/test_kernels_gpu.py tests/test_objective_gpu.py tests/test_rope_gpu.py tests/test_sampling.py tests/test_skinny_gpu.py tests/test_train_gpu.py tests/test_unet_gpu.py -m gpu -q --durations=15 > $o/gputest.log 2>&1; echo "rc=$?" >> $o/gputest.log
tail -3 $o/gputest.log
python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
for b in 256 768; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary > $o/plain_b$b.json 2> $o/plain_b$b.err
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary --grad-checkpoint > $o/ckpt_b$b.json 2> $o/ckpt_b$b.err
done
python bench.py --batch 1536 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-secondary --grad-checkpoint > $o/ckpt_b1536.json 2> $o/ckpt_b1536.err
python bench.py --batch 3072 --steps 6 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary --grad-checkpoint > $o/ckpt_b3072.json 2> $o/ckpt_b3072.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3m1/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["mfma_frac_whole_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
