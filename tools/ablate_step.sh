cd $GRAFT_REPO_ROOT
for a in none copy agg chain_b chain_a mha "agg,chain_b,chain_a,mha" "copy,agg,chain_b,chain_a,mha"; do
  echo -n "ablate $a: "; GD4D_DEV=1 GD4D_ABLATE=$a timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
