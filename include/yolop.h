/* yolop.h - C-ABI of libyolop.so: the MI355X-native YOLOv10 predict hot path.
 *
 * The reference (daisy9542/yolo-puncture) has NO native/FFI layer for this path: its boundary is the Python call
 * `ultralytics.YOLO(path).predict(source, conf=, retina_masks=, device=)` (yolo_seg/app.py:45,49,91;
 * yolo_seg/yolo_with_deva.py:51,226; dev_tools/auto_speed_calc.py:40,62;
 * dev_tools/classify/cls_bbox_dataset_generate.py:48,66). Each entry point below names the part of that call
 * it replaces. The Python facade in yolo-puncture_amd/predictor.py binds these with ctypes.
 *
 * Conventions: every function returns 0 on success, <0 on error; yp_last_error() returns a thread-local
 * message. All device pointers are raw HIP device addresses owned by the CALLER (PyTorch tensors'
 * data_ptr()); the engine owns packed weights + its activation workspace. Work is enqueued on the stream
 * passed in (a hipStream_t cast to void*; NULL = default stream) with no hidden synchronisation.
 * An engine is bound to one device and is not re-entrant: one engine per GPU, one caller thread at a time.
 */
#ifndef YOLOP_H
#define YOLOP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YP_OK 0
#define YP_ERR_ARG -1      /* bad argument / shape                */
#define YP_ERR_STATE -2    /* call order (e.g. forward before finalize) */
#define YP_ERR_HIP -3      /* a HIP runtime call failed           */
#define YP_ERR_WEIGHT -4   /* unknown / missing / mis-shaped weight */

#define YP_BF16 0          /* activations+weights bf16 in HBM, fp32 accumulate (MFMA 16x16x32 bf16) */
#define YP_F32 1           /* everything fp32 (MFMA 16x16x4 f32): strict-parity mode               */

#define YP_TASK_DETECT 0
#define YP_TASK_SEGMENT 1

#define YP_NM 32           /* mask coefficients per detection */
#define YP_MAX_MASKS 480   /* masks per yp_masks / yp_id_mask_resized call (one frame's coefficients stay in LDS); a YP_TASK_SEGMENT engine
                              therefore takes max_det <= 480 - yp_create refuses more (ultralytics' default, and what every caller of the
                              reference leaves in place, is 300: yolo_seg/app.py:91, yolo_seg/yolo_with_deva.py:51) */

typedef struct yp_engine yp_engine;

/* What `YOLO(path)` learns from the checkpoint's yaml: which graph to build. */
typedef struct yp_model_desc {
    int variant; /* ASCII 'n','s','m','b','l','x'                                */
    int nc;      /* number of classes                                             */
    int task;    /* YP_TASK_DETECT | YP_TASK_SEGMENT (v10 trunk + Proto/cv4 head) */
    int dtype;   /* YP_BF16 | YP_F32                                              */
    int max_det; /* 300 (ultralytics default); top-k size of the one-to-one head; <= 1024 (detect), <= YP_MAX_MASKS (segment) */
    int family;  /* YP_FAMILY_V10 (0) | YP_FAMILY_V8 | YP_FAMILY_11: which ultralytics yaml the graph follows. The v8 / 11
                    families (the checkpoints the reference's UI offers, yolo_seg/app.py:218-223) are segment models: task must
                    be YP_TASK_SEGMENT; their head ends in conf filter + NMS (yp_set_nms) instead of the v10 top-k */
} yp_model_desc;

#define YP_FAMILY_V10 0
#define YP_FAMILY_V8 8
#define YP_FAMILY_11 11

const char* yp_last_error(void);

/* -- construction: replaces `YOLO(path)` (yolo_seg/app.py:45). Builds the op graph on the host; touches no GPU
 *    state, so it also works (for inspection) on a box without a GPU. device = HIP ordinal used later. */
int yp_create(const yp_model_desc* desc, int device, yp_engine** out);
int yp_destroy(yp_engine* e);

/* The folded (Conv+BN merged) parameters the graph expects, e.g. "model.0.weight" [32,3,3,3], "model.0.bias" [32]. */
int yp_weight_count(const yp_engine* e);
int yp_weight_info(const yp_engine* e, int i, char* name, int name_cap, int64_t shape[4], int* ndim);

/* Hand one folded fp32 parameter (HOST pointer) to the engine; it is repacked to the kernel layout
 * ([Cout][ky][kx][Cin], K padded) in the engine dtype. Replaces the state-dict load inside YOLO(path). */
int yp_set_weight(yp_engine* e, const char* name, const float* host, const int64_t* shape, int ndim);

/* Verify every parameter was set, upload packed weights to the device. After this the engine can run. */
int yp_finalize(yp_engine* e);

/* -- the forward + NMS-free post-process: replaces predictor inference + `v10postprocess` inside
 *    `.predict(...)` (yolo_seg/app.py:91). in_dev: uint8 NHWC **BGR**, already letterboxed, [B,H,W,3], H and W
 *    multiples of 32. Outputs (device, caller-owned):
 *      det_out   float [B,max_det,6] = x1,y1,x2,y2 (letterboxed-input pixels), score, class; rows sorted by
 *                score descending, ties by (stage-1 rank, class) ascending
 *      idx_out   int32 [B,max_det] anchor index of each row (may be NULL)
 *      coeff_out float [B,max_det,32] mask coefficients of each row (segment task; may be NULL)
 *    If the image has fewer than max_det anchors, k = #anchors rows are valid and the rest are zero. */
int yp_forward(yp_engine* e, const uint8_t* in_dev, int B, int H, int W, float* det_out, int32_t* idx_out,
               float* coeff_out, void* stream);

/* Engine-owned prototype tensor of the last forward (segment task): NHWC [B,Hp,Wp,32] in the engine dtype. */
int yp_proto(const yp_engine* e, const void** proto_dev, int* Hp, int* Wp);

/* -- segmentation tail: replaces ops.process_mask_native / process_mask + the id painting loop of
 *    `auto_segment` (yolo_seg/yolo_with_deva.py:54-86).
 *    image b of the last forward; n rows of coeff [n,32] (device float) and boxes [n,4] (device float):
 *    retina != 0: boxes in ORIGINAL-image pixels, masks at (oh,ow)      (process_mask_native)
 *    retina == 0: boxes in letterboxed-input pixels, masks at (oh,ow)=(H,W) of the forward (process_mask)
 *    masks_out: uint8 [n,oh,ow] in {0,1} (may be NULL)
 *    id_out:    int64 [oh,ow] painted ids (may be NULL): rows in order, later overwrite earlier, rows whose
 *               area < min_area are skipped when suppress_small != 0, ids consecutive over kept rows
 *    kept_out:  int32 [n] id assigned to each row (0 = suppressed) (may be NULL; needs id_out)          */
int yp_masks(yp_engine* e, int b, const float* coeff_dev, const float* boxes_dev, int n, int oh, int ow,
             int retina, uint8_t* masks_out, int64_t* id_out, int32_t* kept_out, int suppress_small,
             int min_area, void* stream);

/* `auto_segment` as the reference normally calls it, with min_side > 0 (yolo_seg/yolo_with_deva.py:45-48,118,140): the frame is
 * shrunk before predict, so the masks come out at (oh,ow) = the shrunk frame and each FLOAT {0,1} mask is resized to the original
 * frame (rh,rw) with torchvision `F.resize` (:71-72: bilinear, align_corners=False, antialias) before the area test
 * `mask.sum() < MIN_AREA_THRESHOLD` (:75, on the resized float mask) and the paint `output_mask[mask > 0.5] = id` (:79).
 * The resize repeats torch's CPU kernel operation by operation (fp32 weights, horizontal pass first, fused multiply-adds), so the
 * many exact 0.5 ties of e.g. a 2:3 upscale fall on the same side. boxes in (oh,ow) pixels; id_out int64 [rh,rw]; kept_out int32 [n].
 * With (rh,rw) == (oh,ow) this is yp_masks(retina=1) with id_out. */
int yp_id_mask_resized(yp_engine* e, int b, const float* coeff_dev, const float* boxes_dev, int n, int oh, int ow, int rh,
                       int rw, int64_t* id_out, int32_t* kept_out, int suppress_small, int min_area, void* stream);

/* `results[0].masks.xy[i]` and `get_coord_min_rect_len(...)` on the device (yolo_seg/app.py:101-103, yolo_seg/utils/mask_tools.py:12-22;
 * [U] Masks.xy = masks2segments(strategy): cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), then "all" (the 8.3.x line the app's
 * YOLO11 weights need: every contour, concatenated) or "largest" (8.0-8.2: the contour with the most points, the first of them on a tie);
 * get_coord_min_rect_len = cv2.minAreaRect of that polygon -> (long side, long/short)). Engine-free. Per mask (one workgroup): bit image
 * of the mask's bounding box in LDS, Moore trace of every blob's outer border, RETR_EXTERNAL by crossing parity (a blob inside a hole of
 * another blob is skipped), run end points, convex hull, rotating calipers. cv2 is not available to pin this against; the definitions are
 * written out in yolo-puncture_amd/hostops.py ("Masks.xy") and restated three ways in the tests: a contour starts at its blob's raster-first
 * pixel and runs down first (a filled rectangle: top-left, bottom-left, bottom-right, top-right); contours are listed bottom-up
 * (descending raster order of the start pixels).
 *    masks_dev uint8 [n,H,W] (non-zero = set)
 *    strategy  YP_CONTOURS_LARGEST or YP_CONTOURS_ALL
 *    pts_out   int32 [n,max_pts,2] (x,y) polygon in mask pixels       count_out int32 [n]: number of points; 0 = empty mask;
 *              -1 = bounding box larger than the LDS image (1280x720 fits), -2 = more than max_pts points / 4096 border starts / 64 outer
 *              borders: use the host path (hostops.mask_polygon)
 *    parts_out int32 [n,parts_cap] or NULL: [0] = contours in the list, [1 .. ] = their point counts in list order (as many as fit)
 *    rect_out  double [n,2] = (long side, short side) of the minimum-area rectangle of the polygon (may be NULL) */
#define YP_CONTOURS_LARGEST 0
#define YP_CONTOURS_ALL 1
int yp_mask_contours(const uint8_t* masks_dev, int n, int H, int W, int strategy, int max_pts, int32_t* pts_out, int32_t* count_out,
                     int32_t* parts_out, int parts_cap, double* rect_out, void* stream);

/* LetterBox on the device (the step before the network inside `.predict`; reference call sites yolo_seg/app.py:86-91,
 * [U] ultralytics LetterBox = cv2.resize INTER_LINEAR + cv2.copyMakeBorder(114)). Engine-free, pure function of its
 * arguments; bit-exact with the fixed-point 8-bit bilinear resize the oracle restates.
 *    src_dev: uint8 [h0,w0,3] (HWC, any channel order)      dst_dev: uint8 [out_h,out_w,3]
 *    the resized image (new_h x new_w; resize skipped when equal to h0 x w0) lands at (top,left), the rest is pad_value.
 *    The geometry is the caller's (LetterBox arithmetic is host integer math: predictor.py / hostops.letterbox_geometry). */
int yp_letterbox(const uint8_t* src_dev, int h0, int w0, uint8_t* dst_dev, int out_h, int out_w, int new_h, int new_w,
                 int top, int left, int pad_value, void* stream);

/* -- U^2-Net-P / U^2-Net: the second per-frame network of the reference's video loop, `unet_predict(unet_model,
 *    cropped_frame)` at yolo_seg/app.py:184. Model: yolo_seg/tasks/models/U2Net.py:424-526 (U2NETP) / :318-420 (U2NET); loader and
 *    post-process: yolo_seg/tasks/unet_segment.py:32-73. variant 'p' = U2NETP, 'f' = U2NET. Weights are handed over folded (conv +
 *    eval-mode BatchNorm -> one weight/bias pair per REBNCONV, named like the reference modules: "stage1.rebnconvin.weight", ...,
 *    "side1.weight", "outconv.weight"), exactly like yp_set_weight.
 *    yp_u2net_forward: bgr_dev uint8 [B,H,W,3] BGR (what cv2 hands the reference; BGR->RGB and /255 of numpy2tensor happen on the device),
 *      prob_out float [B,H,W] = sigmoid(d0), the first of the model's seven outputs (required)
 *      norm_out float [B,H,W] = normPRED(prob) = (p - min) / (max - min) over the whole call (may be NULL)
 *      mask_out uint8 [B,H,W] = norm > 0.5 ? 255 : 0, the array `unet_predict` returns (may be NULL) */
typedef struct yp_u2net yp_u2net;
int yp_u2net_create(int variant, int dtype, int device, yp_u2net** out);
int yp_u2net_destroy(yp_u2net* e);
int yp_u2net_weight_count(const yp_u2net* e);
int yp_u2net_weight_info(const yp_u2net* e, int i, char* name, int name_cap, int64_t shape[4], int* ndim);
int yp_u2net_set_weight(yp_u2net* e, const char* name, const float* host, const int64_t* shape, int ndim);
int yp_u2net_finalize(yp_u2net* e);
int yp_u2net_forward(yp_u2net* e, const uint8_t* bgr_dev, int B, int H, int W, float* prob_out, float* norm_out,
                     uint8_t* mask_out, void* stream);
int yp_u2net_set_graph(yp_u2net* e, int enable);   /* hipGraph replay of the forward (default off: measured slower than eager launches; the first pass of a shape is always eager) */
int yp_u2net_tensor_count(const yp_u2net* e);
int yp_u2net_tensor_info(const yp_u2net* e, int i, char* name, int name_cap, int dims[4] /*B,H,W,C*/);
int yp_u2net_tensor_read(yp_u2net* e, int i, float* host_out);   /* sync copy NHWC -> fp32 host (debug taps) */

/* -- multi-GPU: frames are sharded over one process per GPU; the only data-path collective is ONE all-gather of
 *    the [B/G,max_det,6] detections per batch over RCCL (xGMI). The reference has no multi-GPU path (frames are independent inside
 *    `.predict`, yolo_seg/app.py:85-91). bench.py / parallel.py use torch.distributed's "nccl" backend for it; these entry points
 *    give a host without PyTorch the same collective. librccl.so is resolved at run time (dlopen).
 *      rank 0: yp_comm_unique_id(id) -> ship the 128 bytes to every rank (any channel) -> all ranks: yp_comm_create(id, rank, world, dev)
 *      yp_allgather: recv_dev holds world * bytes_per_rank bytes in rank order; enqueued on `stream`. */
typedef struct yp_comm yp_comm;
int yp_comm_unique_id(void* id128);
int yp_comm_create(const void* id128, int rank, int world, int device, yp_comm** out);
int yp_allgather(yp_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int yp_comm_destroy(yp_comm* c);

/* -- introspection (tests, bench): the planned op list for an input shape; host only. */
int yp_plan(yp_engine* e, int B, int H, int W);            /* (re)build the plan; returns #ops or <0 */
int yp_op_info(const yp_engine* e, int i, char* name, int name_cap, int* kind, double* flops,
               double* bytes);                                 /* algorithmic FLOPs / HBM bytes of op i */
int yp_op_kernel(const yp_engine* e, int i, char* name, int name_cap);   /* device kernel symbol op i launches */
int yp_op_output(const yp_engine* e, int i, int* tensor, int* coff, int* C); /* output channel slice of op i (tensor<0: user buffers) */
/* Input view of op i, and `c_read`: the channels a dense conv packed with padded taps actually reads per pixel (> C for 48- / 80-channel
   inputs: the surplus lanes meet zero weights; their bytes come from the next pixel or, at the very end, from the tensor's zeroed tail). */
int yp_op_input(const yp_engine* e, int i, int* tensor, int* coff, int* C, int* c_read);
/* Fused producer of op i under the current plan: *pre = index of the 1x1 conv that runs as the first stage of op i's kernel (pwsp_kernel:
   1x1 -> depthwise conv / SPPF pool chain, one workgroup per image and channel slice), or -1; *pre_stored = 1 when that kernel also writes
   the 1x1's own output tensor (it has other readers). The parity tests use it to teacher-force both results of such a launch. */
int yp_op_fusion(const yp_engine* e, int i, int* pre, int* pre_stored);
/* Debug stepping for per-op parity tests ("teacher forcing"): run ONE op of the current plan, and overwrite a
 * channel slice of an engine tensor from fp32 host data [B,H,W,C] (converted to the tensor's storage type). */
int yp_run_op(yp_engine* e, int i, const uint8_t* in_dev, float* det_out, int32_t* idx_out, float* coeff_out, void* stream);
int yp_tensor_write(yp_engine* e, int tensor, int coff, int C, const float* host);
int yp_tensor_count(const yp_engine* e);
int yp_tensor_info(const yp_engine* e, int i, char* name, int name_cap, int dims[4] /*B,H,W,C*/, int* is_f32);
int yp_tensor_read(yp_engine* e, int i, float* host_out);  /* sync copy NHWC -> fp32 host (debug taps) */

/* Run the plan op by op with a HIP event pair around every launch on `stream`; ms_out[#ops]. */
int yp_profile(yp_engine* e, const uint8_t* in_dev, int B, int H, int W, float* det_out, int32_t* idx_out,
               float* coeff_out, float* ms_out, int iters, void* stream);

/* NMS heads (families v8 / 11): score threshold and IoU threshold of `ops.non_max_suppression` [U] as `.predict(conf=, iou=)` sets
 * them (defaults 0.25 / 0.7). Rows of yp_forward's det_out are the boxes NMS keeps, best first; may be changed between forwards
 * (the values live in device memory: a captured graph does not go stale). No effect on the v10 family. */
int yp_set_nms(yp_engine* e, float conf, float iou);

/* Enable/disable the plan-time autotuner that picks the conv tile configuration per layer (default on). */
int yp_set_autotune(yp_engine* e, int enable);

/* Tile configurations of the current plan, one id per op (-1 = heuristic / not a tunable conv), as the plan-time autotuner (or an import) left
   them. bf16 results depend on them in the last bit (fp32 summation order), so a job that wants every rank - or every box - to return identical
   detections for a frame tunes once and hands the ids round: rank 0 exports after its first forward, the others import BEFORE theirs
   (yolo_puncture_amd.parallel.sync_tuning does this over torch.distributed; bench.py calls it at N > 1). The reference has no counterpart
   (one process, PyTorch picks its kernels: yolo_seg/app.py:45-50).
   yp_tuning_export: returns the number of ops; fills cfg_out[0..n) when it is non-null (cap >= n). Needs a forward on the current plan.
   yp_tuning_import: plans (B,H,W) and installs the ids for it; every id is validated against this build's configuration tables for its layer
   (all or nothing, YP_ERR_ARG otherwise). The next yp_forward of that shape neither tunes nor reads the tune cache. */
int yp_tuning_export(const yp_engine* e, int32_t* cfg_out, int cap);
int yp_tuning_import(yp_engine* e, int B, int H, int W, const int32_t* cfg, int n);

/* Test hook: force conv tile configuration `cfg` wherever it is valid (-1 = off). Returns the number of configurations. */
int yp_debug_force_conv_cfg(int cfg);
/* Timing ablation for tools (results become wrong): 0 off, 1 conv kernels drop their stores, 2 drop their pixel loads. */
int yp_debug_ablation(int v);
/* Profiling hook: 100 MHz timestamps of the phases of the top-k kernel (image 0's workgroup, last launch):
   [0] start, [1] keys loaded, [2] stage-1 lower bound found, [3] stage-1 select done, [4] stage-2 candidates scanned,
   [5] stage-2 select done, [6] decoded; [7] = stage-2 rounds << 32 | candidates that entered the last round's select. */
int yp_debug_head_clocks(uint64_t* out8);
/* The same for the position kernel of the winners-only head (workgroup 0's first tile, last launch): [0] tile start, [1] position list read,
   [2] first patch plane landed, [3] all channel chunks done, [4] activations stored; [5] = level << 32 | channel chunks of that tile,
   [6] = tiles of the launch (64 positions each), [7] = listed positions of the launch (all levels, all images). */
int yp_debug_head_branch_clocks(uint64_t* out8);
/* Winners-only head (the default for bf16 YOLOv10 engines whose branch width is 64 / 32 channels; YOLOP_DENSE_HEAD=1 at yp_create keeps
   every branch dense): the box branch `one2one_cv2.*` - and the coefficient branch `cv4.*` of the seg head - are evaluated only for the
   max_det anchors the class scores select, which are the only ones whose boxes the reference's v10postprocess gathers ([U] SURVEY A.6;
   inside `.predict`, yolo_seg/app.py:91): the first 3x3 once per distinct position of the winners' 3x3 neighbourhoods, the second 3x3 and
   the 1x1 at the winners. Same inputs, weights and rounding points as the dense convolutions, so the same numbers up to fp32 summation
   order. This hook copies what the last forward left: sel_host int32 [B][512] (stage-1 winners, anchor ids by rank),
   box_host float [B][max_det][64] and coeff_host float [B][max_det][32] (rows by rank; any may be NULL).
   Returns bit 0: box rows are winners-only, bit 1: coefficient rows are; 0 = dense head. */
int yp_debug_head_winners(yp_engine* e, int32_t* sel_host, float* box_host, float* coeff_host);
/* phase stamps of yp_mask_contours (mask 0): [0..6] 100-MHz ticks at box / bit image / candidates / trace / emit / hull / end, [8] candidates,
   [9] points of the winning contour, [10], [11] bounding box width, height */
int yp_debug_contour_clocks(uint64_t* out12);
/* phase stamps (core clock) of workgroup 0 in the last pwsp_kernel launch: [0] start, [1] prologue issued, [2] GEMM done, [3] epilogue done,
   [4] behind the barrier, [5] spatial stage done, [8 + g] top of k-step g (g < 16) */
int yp_debug_pwsp_clocks(uint64_t* out32);

/* Where the tile configurations of the current plan came from: 0 = this process's tuner (or the heuristics, with autotuning off),
 * 1 = a YOLOP_TUNE_CACHE file, 2 = a table packaged beside the library (tune_tables/tt_<key>.txt: the best of several tunings of that
 * shape; used when YOLOP_TUNE_CACHE is unset; YOLOP_NO_TUNE_TABLES=1 ignores them). */
int yp_tuning_source(const yp_engine* e);

/* Host-only self-check of the executor for the current plan (parameter blocks, kernel symbols, tune-cache round trip, lane
 * schedule invariants). Needs no GPU; returns the number of scheduled launches or <0. Used by the CPU sanitizer build. */
int yp_debug_host_selftest(yp_engine* e);
/* What the last capture built: out[0] captures + instantiations so far, [1] graph nodes, [2] graph edges, [3] edges into the first node
   of every op as the lane schedule prescribes them (for a chain: nodes - 1; [2] - [3] = the inner chains of ops that launch several
   kernels), [4] lanes that launched, [5] 1 while an executable exists. */
int yp_debug_graph_info(const yp_engine* e, int64_t* out6);
/* Winners-only head, last forward: out[0..3) distinct positions listed per level (P3, P4, P5; all images), out[3..6) winners per level.
   Returns 0 for a dense head (out zeroed). Synchronises the device. */
int yp_debug_head_positions(yp_engine* e, int64_t* out6);
/* Profiling hook: enqueue a one-thread kernel named op_marker_kernel (tools/op_traffic.py brackets the launches of one op with it so that
   per-dispatch counter rows can be attributed to ops). */
int yp_debug_marker(void* stream);

/* hipGraph capture + replay of the forward: 0 = eager launches on the caller's stream (the library default), 1 = one hipGraph whose
 * independent head branches are parallel branches of the graph, 2 = the graph as one chain (what a ring of several engines uses; A/B),
 * 3 = auto: per input shape, whichever of 0 and 1 a one-off timing inside the first yp_forward of that shape finds faster on this box
 * (what predictor.py uses: one frame per call - yolo_seg/app.py:85-91 - runs eagerly, batches replay). The first forward of every
 * input shape runs once eagerly inside yp_forward before anything is captured.
 * Stream contract: a replay is launched on the CALLER's stream (`stream` of yp_forward); a caller on the legacy NULL stream gets the
 * engine's own stream ordered in between by an event pair. Forwards of one engine never overlap: a call on another stream than the
 * engine's previous call first waits (on the device) for that call's completion event. The graph is specialised on the plan, the input
 * pointer and the output pointers; a caller that keeps them from call to call never re-captures (yp_debug_graph_info counts captures). */
int yp_set_graph(yp_engine* e, int enable);

#ifdef __cplusplus
}
#endif
#endif /* YOLOP_H */
