# renderer backward, timing-only ablations inside the real step (tools/exp/patches/render_bwd2_ablations.patch builds the variants)
for v in "" NOAUX NOSTAGE NOADJ NOTAPS NOPASSB NOPASSA; do
  if [ -z "$v" ]; then FWD_ONLY=0 python tools/exp/render_ablate.py; else FWD_ONLY=0 SPAIR_HIP_LIB=build/libspair_rb_$v.so python tools/exp/render_ablate.py; fi
done 2>&1 | grep render_
