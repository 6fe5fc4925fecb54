for op in model.6.cv2 model.13.cv1 model.13.cv2 model.19.cv1 model.19.cv2; do
  for c in 800 801 802 803; do
    timeout -k 5 100 python3 tools/op_bench.py --op $op --cfg $c --iters 30 2>/dev/null | tail -1
  done
done
