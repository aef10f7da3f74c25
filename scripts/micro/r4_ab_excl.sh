cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in lib lib_excl; do
  export IDE3D_HIP_LIB=$PWD/ide-3d_amd/$L/libide3d_hip.so
  echo "== $L"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-roofline-extra --no-dropin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['by_conv_arithmetic'])"
done
