for v in "" abl_NOGLST abl_NOGXY; do
  if [ -z "$v" ]; then BWD=1 python tools/exp/chain_ablate.py; else BWD=1 SPAIR_HIP_LIB=build/libspair_$v.so python tools/exp/chain_ablate.py; fi
done 2>&1 | grep cells
SPAIR_HIP_LIB=build/libspair_abl_NOGLST.so python tools/chain_stamps.py 2>&1 | sed -n 2,19p
SPAIR_HIP_LIB=build/libspair_abl_NOGXY.so python tools/chain_stamps.py 2>&1 | sed -n 20,42p
