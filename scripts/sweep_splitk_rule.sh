for rule in "64,256,4,8" "128,256,4,8" "128,256,8,16" "192,384,8,16" "256,512,8,16" "128,512,4,16" "96,256,4,8"; do
  echo "== rule $rule"; RASS_GEMM_SPLITK_RULE=$rule python scripts/probe_encoder_shapes.py 8x12 16x12 24x12 32x12 48x12 64x12 32x32 48x32 4x512 6x512 8x512 2>&1 | grep tokens
done
