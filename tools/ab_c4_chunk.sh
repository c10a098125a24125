for i in 1 2 3; do
  for c in 1 0; do
    v=$(TF_C4_CHUNK=$c python bench.py --images 4 --latent 96 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-config5 --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $i chunk=$c: $v ms/step"
  done
done
