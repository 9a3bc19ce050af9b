for lib in "" ablate2 ablate4 ablate6; do
  if [ -z "$lib" ]; then echo "== full"; python tools/hit_scaling.py 2>&1 | grep -E "^(plane|sphere |lens|sphere&)" ;
  else echo "== $lib"; PRT_LIB=$GRAFT_REPO_ROOT/pyrayt_amd/csrc/libprt_hip_$lib.so python tools/hit_scaling.py 2>&1 | grep -E "^(plane|sphere |lens|sphere&)"; fi
done
