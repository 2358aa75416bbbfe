for R in 2 3 4; do
  touch bayesianinferencedl_amd/csrc/fom_band_adjoint.hip
  FINROM_EXTRA_FLAGS=-DADJ_RING=$R python -m bayesianinferencedl_amd._build > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== ADJ_RING=$R"
  timeout -k 10 200 python tools/fom_grad_bench.py five 2>&1 | tail -1
  timeout -k 10 300 python tools/fom_grad_bench.py field 20 20000 2>&1 | tail -1
done
