for op in model.4.cv1 model.4.cv2 model.16.cv1 model.16.cv2 model.6.cv1 model.6.cv2 model.7.cv1 model.13.cv2 model.19.cv1 model.20.cv1; do
  for c in 1100 1102 1200 1201; do
    timeout -k 5 100 python3 tools/op_bench.py --op $op --cfg $c --iters 30 2>/dev/null | tail -1
  done
done
