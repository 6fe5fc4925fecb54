for op in model.4.cv1 model.4.cv2 model.16.cv1 model.16.cv2 model.6.cv1 model.6.cv2 model.19.cv1 model.20.cv1; do
  for c in -1 1100 1101 1102; do
    timeout -k 5 100 python3 tools/op_bench.py --op $op --cfg $c --iters 30 2>/dev/null | tail -1
  done
done
for op in model.23.one2one_cv3.0.2 model.23.one2one_cv3.1.2; do timeout -k 5 100 python3 tools/op_bench.py --op $op --iters 30 2>/dev/null | tail -1; done
