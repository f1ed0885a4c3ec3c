/*
 * eigen_engine.h -- C ABI of the MI355X (gfx950) fitness-evaluation engine for EIGen.
 *
 * Drop-in boundary for ONE hot path of LanaSina/evolutionary_illusion_generator: per genome
 *   CPPN render -> PredNet roll-out (20 repeats + self-fed frames) -> Lucas-Kanade flow -> motion score.
 * The reference is pure Python and has no FFI; each entry point below replaces the Python-level call the
 * reference makes at the cited file:line, and is what a ctypes binding on the reference side would bind
 * (stub shown in INTEGRATION.md).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative eigen_status; eigen_last_error() gives the text
 *     (thread-local);  no exception crosses the ABI.
 *   - pointers named h_* are HOST memory, d_* are DEVICE (HIP) memory on the engine's device.  The caller
 *     owns all of them; the engine owns only its handle and its internal device workspaces.
 *   - all device work is enqueued on the caller-supplied hipStream_t (passed as void*; NULL = default
 *     stream).  Functions that return host results (h_* outputs) synchronise that stream before returning.
 *   - a handle is not thread-safe; use one handle per process/rank (one rank per GPU).
 *   - images are PLANAR uint8 [B][C][H][W] (C = 1 gray or 3 RGB), the CHW form the reference feeds PredNet.
 */
#ifndef EIGEN_ENGINE_H
#define EIGEN_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIGEN_MAX_LAYERS 8
#define EIGEN_ABI_VERSION 4 /* 2: eigen_config grew (flow_method, fb_*), eigen_debug_dense_flow, gradient = 2; 3: eigen_gate_order; 4: eigen_winograd_mask */

typedef enum {
    EIGEN_OK = 0,
    EIGEN_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    EIGEN_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
    EIGEN_ERR_STATE = -3,     /* call order: weights / grid not set */
    EIGEN_ERR_CAPACITY = -4   /* batch, genome or feature count exceeds what the handle was created for */
} eigen_status;

/* StructureType of the reference (generate_illusion.py:25-29, fitness_calculator.py:10-14) */
typedef enum { EIGEN_BANDS = 0, EIGEN_CIRCLES = 1, EIGEN_FREE = 2, EIGEN_CIRCLES_FREE = 3 } eigen_structure;
/* eigen_score only: inside_outside_score(vectors, width, height) (fitness_calculator.py:219-304) on ALL the given vectors.
 * The reference reaches it only through an else branch that raises NameError (generate_illusion.py:606-607,
 * fitness_calculator.py:545-546), so no structure selects it on the population path. */
#define EIGEN_SCORE_INSIDE_OUTSIDE 4

/* Which two frames Lucas-Kanade compares:
 *   POPULATION: prediction after step n_repeat -> first extension frame   (generate_illusion.py:543-550)
 *   SINGLE    : the input image -> second extension frame                 (fitness_calculator.py:493-498) */
typedef enum { EIGEN_PAIR_POPULATION = 0, EIGEN_PAIR_SINGLE = 1 } eigen_pairing;

/* CPPN activation ids (PyTorch-NEAT activations.py; neat_configs/ circles.txt:12 lists the options in use) */
typedef enum {
    EIGEN_ACT_SIGMOID = 0, /* 1/(1+exp(-5x)) */
    EIGEN_ACT_TANH = 1,    /* tanh(2.5x)     */
    EIGEN_ACT_ABS = 2,
    EIGEN_ACT_GAUSS = 3,   /* exp(-5x^2)     */
    EIGEN_ACT_IDENTITY = 4,
    EIGEN_ACT_SIN = 5,
    EIGEN_ACT_RELU = 6
} eigen_activation;

typedef struct {
    int32_t device;                        /* HIP device ordinal */
    int32_t width, height;                 /* image size; both divisible by 2^(n_layers-1) */
    int32_t n_layers;
    int32_t channels[EIGEN_MAX_LAYERS];    /* PredNet channels, channels[0] = c_dim (1 or 3) */
    int32_t max_batch;                     /* genomes evaluated per device batch (workspace sizing) */
    /* constants hard-coded in the reference (generate_illusion.py:482,531; fitness_calculator.py:470-471) */
    int32_t n_repeat;                      /* 20 */
    int32_t n_ext;                         /* 2  */
    int32_t requant_feedback;              /* 0: extension frames feed the float prediction back (upstream) */
    /* Optical_Flow_Analyzer / OpenCV parameters (goodFeaturesToTrack, calcOpticalFlowPyrLK) */
    int32_t lk_max_corners;                /* 100  (<= 128) */
    int32_t lk_block_size;                 /* 7    */
    int32_t lk_win;                        /* 15   */
    int32_t lk_max_level;                  /* 2    */
    int32_t lk_max_iter;                   /* 10   */
    int32_t flow_method;                   /* EIGEN_FLOW_LK (0, what the reference calls) or EIGEN_FLOW_FARNEBACK (1) */
    double lk_quality_level;               /* 0.3  */
    double lk_min_distance;                /* 7    */
    double lk_epsilon;                     /* 0.03 */
    double lk_min_eig_thr;                 /* 1e-4 */
    /* Dense flow after Farneback, cv::calcOpticalFlowFarneback with the parameters of OpenCV's dense-flow tutorial
     * (pyr_scale is fixed at 0.5).  The reference never calls a dense-flow routine (its only flow call is lucas_kanade,
     * generate_illusion.py:549-550); the option answers the north star's "Farneback/Lucas-Kanade flow".  The dense field is
     * sampled every fb_step pixels from fb_step/2 (the grid of OpenCV's samples/python/opt_flow.py; the step grows by
     * multiples of fb_step until the grid fits lk_max_corners vectors) into the same [x, y, dx, dy] vectors. */
    int32_t fb_levels;                     /* 3  (levels on top of the full resolution; fewer if a level would be < 32 px) */
    int32_t fb_winsize;                    /* 15 (odd, <= 33) */
    int32_t fb_iterations;                 /* 3  */
    int32_t fb_poly_n;                     /* 5  (<= 7) */
    int32_t fb_step;                       /* 16 */
    int32_t reserved1;
    double fb_poly_sigma;                  /* 1.2 */
} eigen_config;

typedef enum { EIGEN_FLOW_LK = 0, EIGEN_FLOW_FARNEBACK = 1 } eigen_flow_method;

/* A batch of CPPN genomes, flattened by the host (create_cppn semantics, generate_illusion.py:384-389):
 * per genome g, nodes node_off[g]..node_off[g+1]-1 are in topological order; node n has incoming edges
 * edge_off[n]..edge_off[n+1]-1 (global edge index), summed left to right in that order;
 * value(n) = act(response * sum_k(weight_k * value(src_k)) + bias).
 * edge_src >= 0 : index of a node of the same genome (relative to node_off[g]);
 * edge_src <  0 : leaf plane  -1 -> plane 0 (x), -2 -> plane 1 (y), ...; -(n_planes+1) -> the constant 1.0.
 * out_node[g*c_out + c]: node (relative index) rendered into channel c. */
typedef struct {
    int32_t n_genomes;
    int32_t c_out;              /* outputs rendered per genome: c_dim, or 1 when gradient == 0 */
    const int32_t* node_off;    /* [n_genomes + 1] */
    const int32_t* edge_off;    /* [total_nodes + 1] */
    const uint8_t* node_act;    /* [total_nodes] eigen_activation */
    const double* node_bias;    /* [total_nodes] */
    const double* node_resp;    /* [total_nodes] */
    const int32_t* edge_src;    /* [total_edges] */
    const double* edge_w;       /* [total_edges] */
    const int32_t* out_node;    /* [n_genomes * c_out] */
} eigen_genome_batch;

typedef struct eigen_engine eigen_engine;

int eigen_abi_version(void);
const char* eigen_last_error(void);

/* Defaults = the constants of the reference + OpenCV tutorial LK parameters; fills every field but
 * device/width/height/n_layers/channels/max_batch. */
void eigen_config_defaults(eigen_config* cfg);

int eigen_create(const eigen_config* cfg, eigen_engine** out);
int eigen_destroy(eigen_engine* e);

/* Replaces `serializers.load_npz(initmodel, model)` inside test_prednet (generate_illusion.py:533).
 * h_tensors: host float32 tensors in the order of evolutionary_illusion_generator_amd.weights.tensor_names()
 * (per layer: [ConvA W,b (l>0)], ConvP W,b, per gate i,f,c,o: x0 W, [x1 W], h W, h b; peepholes c_i,c_f,c_o). */
int eigen_set_prednet_weights(eigen_engine* e, const float* const* h_tensors, int32_t n_tensors);

/* Replaces the per-generation create_grid result (generate_illusion.py:501) as consumed by
 * get_image_from_cppn (:375-378): n_planes float64 planes of H*W values; background where plane 0 == -1. */
int eigen_set_grid(eigen_engine* e, const double* h_planes, int32_t n_planes);

/* Replaces get_image_from_cppn (generate_illusion.py:372-460) for a whole batch.
 * d_images: uint8 [n_genomes][c_dim][H][W].  bg: 1 white / 0 black.  gradient: 1 / 0 as in the reference's
 * get_image_from_cppn (generate_illusion.py:372-460); 2 = get_equilum_image_from_cppn (:333-367, c_dim 3): the three
 * output nodes are h, s, v, converted per pixel as colorsys.hsv_to_rgb does, bg applied to h, s and v beforehand. */
int eigen_render_cppn(eigen_engine* e, const eigen_genome_batch* h_genomes, int32_t bg, int32_t gradient,
                      uint8_t* d_images, void* stream);

/* Replaces the node calls `node_func(x=inp_x, y=inp_y)` on the objects create_cppn returns (PyTorch-NEAT
 * Node.__call__; generate_illusion.py:395, 406, 443) for a batch: the raw float64 value of every output node
 * at every pixel of the planes given to eigen_set_grid, no background fill and no quantisation.
 * d_nodes: float64 [n_genomes][c_out][H*W].  Used by the import shim pytorch_neat.pytorch_neat.cppn. */
int eigen_eval_cppn_nodes(eigen_engine* e, const eigen_genome_batch* h_genomes, double* d_nodes, void* stream);

/* Replaces test_prednet (generate_illusion.py:533-537, fitness_calculator.py:487-491) for a batch: state
 * reset, n_repeat steps on the constant frame, then extension steps feeding the prediction back; runs
 * n_steps <= n_repeat + n_ext steps in total.  The quantised prediction of every step t >= first_out_step
 * is written to d_frames[b][t - first_out_step] (uint8 [B][n_steps - first_out_step][C][H][W]). */
int eigen_prednet_rollout(eigen_engine* e, const uint8_t* d_images, int32_t batch, int32_t n_steps,
                          int32_t first_out_step, uint8_t* d_frames, void* stream);

/* Replaces lucas_kanade (generate_illusion.py:549-550, fitness_calculator.py:498) for a batch of pairs.
 * d_img0/d_img1: uint8 planar images, image b at d_imgX + b * strideX (bytes).
 * d_vectors: float [batch][lk_max_corners][4] rows [x, y, dx, dy]; d_counts: int32 [batch]. */
int eigen_flow(eigen_engine* e, const uint8_t* d_img0, int64_t stride0, const uint8_t* d_img1, int64_t stride1,
               int32_t batch, float* d_vectors, int32_t* d_counts, void* stream);

/* Replaces the scoring block of get_fitnesses_neat (generate_illusion.py:559-616; scorers
 * fitness_calculator.py:18-215).  count == 0 -> the sentinel [[0,0,-1000,0]] -> fitness 0.
 * width/height: the image size the scorers are given (w, h); 0 = the engine's own size.
 * structure EIGEN_SCORE_INSIDE_OUTSIDE: inside_outside_score of the raw vectors (no filter, no sentinel). */
int eigen_score(eigen_engine* e, int32_t structure, int32_t width, int32_t height, const float* d_vectors,
                const int32_t* d_counts, int32_t batch, double* d_fitness, void* stream);

/* The whole path for one batch of genomes: render -> roll-out -> flow -> score.  h_fitness: host double[B].
 * This is what eval_genomes / get_fitnesses_neat (generate_illusion.py:478-673, 692-694) calls per shard. */
int eigen_eval_population(eigen_engine* e, const eigen_genome_batch* h_genomes, int32_t structure, int32_t bg,
                          int32_t gradient, int32_t pairing, double* h_fitness, void* stream);

/* Same, for ready-made images (single-image API, fitness_calculator.py:468-548). */
int eigen_eval_images(eigen_engine* e, const uint8_t* d_images, int32_t batch, int32_t structure,
                      int32_t pairing, double* h_fitness, float* h_vectors, int32_t* h_counts, void* stream);

/* ---- kernel-level entry points used by the parity tests and bench.py's roofline leg ---- */

/* One fused 3x3 convolution chain on the MFMA kernel: out[b][o][y][x] = sum over sources/channels/taps, raw
 * accumulators (no bias).  d_src[i]: float [batch][cin[i]][H>>up[i]][W>>up[i]]; h_w[i]: host float
 * [cout][cin[i]][3][3].  d_out: float [batch][cout][H][W]. */
int eigen_test_conv(eigen_engine* e, int32_t n_src, const float* const* d_src, const int32_t* cin,
                    const int32_t* up, const float* const* h_w, int32_t cout, int32_t H, int32_t W,
                    int32_t batch, float* d_out, void* stream);

/* eigen_test_conv's launch repeated `iters` times between two HIP events on the launch stream; *h_ms = average
 * kernel duration in milliseconds (measurement hook of scripts/ and bench.py). */
int eigen_time_conv(eigen_engine* e, int32_t n_src, const float* const* d_src, const int32_t* cin, const int32_t* up,
                    const float* const* h_w, int32_t cout, int32_t H, int32_t W, int32_t batch, float* d_out,
                    int32_t iters, double* h_ms, void* stream);

/* Host-side helper (no device work): flattens a batch of NEAT genomes, given as plain arrays, into the arrays behind
 * eigen_genome_batch -- the graph part of what PyTorch-NEAT's create_cppn does for the reference
 * (generate_illusion.py:384-389: which connections are expressed, evaluation order, constant nodes).  It is the C twin of
 * evolutionary_illusion_generator_amd/genome.py: _flatten_lists (same output, element for element; tests/test_host_logic.py).
 *   per genome g: connections conn_off[g]..conn_off[g+1]-1 in genome.connections order (in-key, out-key, weight, enabled);
 *   nodes node_off[g]..node_off[g+1]-1 (key, activation id or 255 if unknown, 1 if aggregation == "sum", bias, response).
 *   input_keys / output_keys: config.genome_config; leaves are numbered in input_keys order.
 * Outputs (caller-allocated, capacities cap_nodes / cap_edges): o_node_off [G+1], o_edge_off [nodes+1], o_act, o_bias,
 * o_resp, o_edge_src, o_edge_w, o_out_node [G][n_outputs].  o_status[g]: 0 done; 1 = a node whose inputs are ALL constants
 * would have to be folded with numpy's float32 activations -- the caller flattens that genome itself (its segment is
 * empty); 2 = the genome is invalid (cycle, unknown activation, unsupported aggregation, missing node): the caller's own
 * code raises the matching exception.  Returns EIGEN_OK, or EIGEN_ERR_CAPACITY if an output array is too small. */
int eigen_flatten_genomes(int32_t n_genomes, int32_t n_inputs, const int32_t* input_keys, int32_t n_outputs,
                          const int32_t* output_keys, const int32_t* conn_off, const int32_t* conn_in,
                          const int32_t* conn_out, const double* conn_w, const uint8_t* conn_enabled,
                          const int32_t* node_off, const int32_t* node_key, const uint8_t* node_act,
                          const uint8_t* node_agg_sum, const double* node_bias, const double* node_resp,
                          int32_t cap_nodes, int32_t cap_edges, int32_t* o_node_off, int32_t* o_edge_off, uint8_t* o_act,
                          double* o_bias, double* o_resp, int32_t* o_edge_src, double* o_edge_w, int32_t* o_out_node,
                          uint8_t* o_status);

/* Element-wise order of the ConvLSTM gate epilogue this library was compiled with (csrc/conv_mfma.h: EIG_GATE_ORDER):
 * 1 = chainer_prednet's ConvLSTM.__call__ where it is knowable (rounded peephole products, sigmoid = tanh(x/2)/2 + 1/2, un-fused
 * cell update), 0 = rounds 1-3.  The CPU oracle (oracle/eig_oracle.c: eig_oracle_gate_order) must report the same value. */
int eigen_gate_order(void);

/* The EFFECTIVE operator-form mask of this process (csrc/eigen_engine.hip: EIGEN_WINOGRAD with EIGEN_WINO_FUSEUP=0 folded in as a cleared bit 24): which 3x3
 * convolutions run in which canonical summation order.  Every rank of a multi-GPU run must report the same value -- a population scored
 * under two orders still looks valid (bench.py gathers it).  The CPU oracle's oracle.wino_mask_default() states the same rule. */
int eigen_winograd_mask(void);

/* Deterministic fp32 exp / sigmoid / tanh used by the gate epilogue. */
int eigen_test_det_math(eigen_engine* e, const float* d_x, int32_t n, float* d_exp, float* d_sig, float* d_tanh,
                        void* stream);

/* Per-stage timing of the last eigen_eval_* call, milliseconds, measured with HIP events on the stream:
 * [0] render  [1] PredNet roll-out  [2] flow  [3] score  [4] PredNet conv kernels only (sum)
 * [5] number of conv kernel launches in the roll-out. */
int eigen_get_timings(eigen_engine* e, double* h_ms6);

/* Per-op profile of the roll-out convolutions.  enable=1 brackets every conv launch with HIP events recorded on the
 * launch stream (and waits for each).  h_out rows (8 doubles each, at most max_ops rows):
 * [layer, epilogue (1 LSTM, 2 ConvA, 3 ConvP, 4 packed LSTM, 5 / 6 the 2x2-form pass over the ConvLSTM's unpooled source (6: all four parity classes in one block); +32 for an operator in its Winograd F(4x4, 3x3) form (csrc/conv_wino4.h; FLOPs then count its 36 multiply-adds per channel and 4x4 outputs -- 25 for an unpooled source); +16 for the step-0 operators, which skip the sources
 *  that are identically zero after reset_state()), NI, TW, launches, total_ms, FLOPs per launch per image (2 x the
 *  multiply-accumulates executed), n_nblk].  At most 6 rows per layer.  reset=1 clears the accumulators after reading. */
int eigen_conv_profile(eigen_engine* e, int32_t enable, int32_t reset, double* h_out, int32_t max_ops, int32_t* n_ops);

/* Stage-level read-back for the parity tests: the dense field of the last eigen_flow call of an engine created with
 * flow_method = EIGEN_FLOW_FARNEBACK, float [batch][2][H][W] (dx plane, dy plane). */
int eigen_debug_dense_flow(eigen_engine* e, int32_t batch, float* h_flow, void* stream);

/* Stage-level read-back for the parity tests: corners / tracked points / status of the last eigen_flow call. */
int eigen_debug_corners(eigen_engine* e, int32_t batch, float* h_corners, int32_t* h_ncorners, float* h_next,
                        uint8_t* h_status, void* stream);

/* Algorithmic FLOPs (2 x MACs of the 3x3 convolutions) of one PredNet step for one genome. */
double eigen_prednet_flops_per_step(const eigen_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* EIGEN_ENGINE_H */
