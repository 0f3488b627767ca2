// engine.hip — host side of libmolnextr_hip.so: weight packing, workspace, encode/decode/edges orchestration,
// hipGraph replay of the decode step. Implements include/molnextr_hip.h.
#include "../../include/molnextr_hip.h"

#include <hip/hip_runtime.h>

#include <time.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "dec_types.h"
#include "kernels.h"
#include "kvq.h"

using namespace mnx;

namespace {

thread_local std::string g_create_error;   // message of the calling thread's last failed mnx_create

// One GEMM weight in its operand form: a single 16-bit (or fp32) plane, or — split modes — a hi plane with the lo plane
// `lo` elements behind it, both holding 2^k * W (k per matrix, fp16 split only); oscale = 2^-k.
struct W16 {
    void* p = nullptr;
    size_t lo = 0;
    float oscale = 1.f;
};
struct BlockW {
    float *ln1_g, *ln1_b, *table, *qkv_b, *proj_b, *ln2_g, *ln2_b, *fc1_b, *fc2_b;
    W16 qkv_w, proj_w, fc1_w, fc2_w;
};
struct StageW {
    std::vector<BlockW> blocks;
    float *m_g = nullptr, *m_b = nullptr;
    W16 m_w;
    int C = 0, heads = 0;
};
// op classes of the split modes (mnx_set_split_terms)
enum { SPL_QKV = 1, SPL_ATTN = 2, SPL_PROJ = 4, SPL_FC1 = 8, SPL_FC2 = 16, SPL_MERGE = 32, SPL_ALL = 63 };
struct GraphKey {
    int slots, rows, trace, forced, tile, branches;
    bool operator<(const GraphKey& o) const {
        return std::tie(slots, rows, trace, forced, tile, branches) < std::tie(o.slots, o.rows, o.trace, o.forced, o.tile, o.branches);
    }
};
constexpr int MAX_TICK_BRANCHES = 8;

}  // namespace

#ifndef MNX_PLANE_SKEW
#define MNX_PLANE_SKEW 4352
#endif
static constexpr size_t PLANE_SKEW = MNX_PLANE_SKEW;   // elements (8704 bytes; 16-byte aligned for the LDS-DMA)

struct mnx_engine {
    mnx_config cfg;
    int device = 0;
    std::string err;
    std::vector<void*> allocs;
    size_t bytes = 0;
    // encoder
    float *pe_wt = nullptr, *pe_b = nullptr, *pe_g = nullptr, *pe_beta = nullptr, *fn_g = nullptr, *fn_b = nullptr;
    std::vector<StageW> stages;
    float *xa = nullptr, *xb = nullptr;              // fp32 residual stream ping-pong
    void *xn16 = nullptr, *qkv16 = nullptr, *attn16 = nullptr, *h16 = nullptr;
    size_t xn_lo = 0, qkv_lo = 0, attn_lo = 0, h_lo = 0;   // split modes: element offset of each buffer's lo plane
    int dt = 0;                                             // kernels.h MNX_DT_* the encoder kernels run (FP16X3M -> MNX_DT_F16X3)
    int split_mask = SPL_ALL;                               // op classes evaluated with their full term count (others: hi.hi only)
    // per stage: op classes whose full term count is TWO (ah.wh + ah.wl) in the blocks [two_first, two_last] of the stage (the
    // patch-merging reduction counts as the stage's last block): FP16X3M, mnx_set_op_terms
    int two_mask[4] = {0, 0, 0, 0};
    int two_first[4] = {0, 0, 0, 0}, two_last[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30};
    int* enc_flag = nullptr;                                // device: set when the final LayerNorm sees a non-finite row
    float* zero_bias = nullptr;                             // [2 * widest C] zeros: the bias of the patch-merging reductions
    int zero_bias_n = 0;
    int tap_item = -1;
    float* tap_dst = nullptr;
    // decoder
    DecWeights dw{};
    DecBuffers db{};
    float* out_trace = nullptr;
    int* forced_ids = nullptr;  // [32, max_len] teacher-forcing ids of mnx_decode_forced (lazy; test aid)
    BeamBuffers beam{};        // allocated lazily on the first mnx_decode_beam
    int* prep_bbox = nullptr;  // scratch of mnx_preprocess
    float* beam_hidden = nullptr;   // mnx_predict_beam: [32, max_len, dec_dim] decoder outputs of the winners (lazy)
    int* host_flag = nullptr;  // pinned: [2][1 + MAX_CHUNKS] poll snapshots + slot lists
    std::map<GraphKey, hipGraphExec_t> graphs;
    // continuous-batching pipeline (mnx_predict)
    hipStream_t enc_stream = nullptr;
    hipEvent_t ev_order = nullptr;
    float* feat_ring[2] = {nullptr, nullptr};
    hipEvent_t ev_enc_done[2] = {nullptr, nullptr}, ev_feat_free[2] = {nullptr, nullptr}, ev_poll[2] = {nullptr, nullptr};
    int* slot_lists = nullptr;          // device [MAX_CHUNKS][32]
    TokenClasses* tc_dev = nullptr;
    bool have_tc = false;
    int n_chunk_bufs = 0;
    bool use_graph = true;
    // greedy ticks of up to dec_fused_max rows run as three launches per layer (dec_fused.hip): dec_tile rows per workgroup in
    // the two attention stages (256 threads per row), dec_tile_ff rows in the feed-forward stage; larger ticks keep the
    // 8-launches-per-layer kernels of decoder.hip (knobs MNX_DEC_TILE, MNX_DEC_TILE_FF, MNX_DEC_FUSED_MAX)
    int dec_tile = -1, dec_tile_ff = 4, dec_fused_max = 128;
    // ticks of more than dec_fused_max and up to dec_mid_max rows run the MID form (dec_fused.hip: dec_fa cut into a 16-row
    // linear launch and an attention launch, 4 launches per layer; bit-identical to the fused form, so that the capacity the
    // host happens to pick — it follows poll timing — is invisible in the results up to dec_mid_max rows); beyond that the
    // 8-launches-per-layer kernels of decoder.hip, whose 32-row linears move the fewest bytes per row (MNX_DEC_MID_MAX;
    // 4096 = every capacity: bit-reproducible jobs of any size, slower at >= 1024 rows)
    int dec_mid_max = 0;
    int dec_xcd = 0;           // fused tick: row tiles pinned to XCDs so that a row's partial planes stay in one L2 (MNX_DEC_XCD)
    // a tick of more than dec_branch_rows rows can be enqueued as up to dec_branch_max BRANCHES of rows on parallel branches
    // of the tick graph (rows are independent through the whole stack). OFF by default (0): measured, the branches of a
    // hipGraph do overlap but every launch gets slower and the fork / join costs more than the overlap buys — 192 rows as
    // 2 x 96: 411 us against 374, 640 rows as 4 x 160: 1552 against 572. Kept as a tested knob
    // (MNX_DEC_BRANCH_ROWS, MNX_DEC_BRANCH_MAX): the row_base plumbing costs nothing.
    int dec_branch_rows = 0, dec_branch_max = 4;
    hipStream_t tick_streams[MAX_TICK_BRANCHES] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_TICK_BRANCHES] = {};   // dec_tile -1: 2 rows per workgroup up to 64 rows of capacity, 4 beyond
    hipStream_t own_stream = nullptr;   // used when the caller passes the legacy null stream (not capturable)
    // profiling (bench aid)
    bool profiling = false;
    int prof_stride = 1;       // bracket the GEMMs of every prof_stride-th mnx_encode call ...
    int prof_calls = 0;        // ... counted since mnx_profile_enable
    int prof_groups = 0;       // encode calls bracketed so far (capped: the event pool stays small)
    int prof_max_groups = 4;
    struct Ev { hipEvent_t a, b; double work; int kind; };   // kind 0 / 4 GEMM (work = FLOP; 4 = block Linears with C >= 512), 1 LayerNorm, 2 window attention, 3 patch embed (work = algorithmic HBM bytes)
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
};

namespace {

#define HIPCHK(h, expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                 \
            return MNX_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)

struct Packer {
    mnx_engine* h;
    std::unordered_map<std::string, const mnx_weight_desc*> by_name;
    std::vector<std::string> problems;
    float* staging = nullptr;
    size_t staging_elems = 0;

    const mnx_weight_desc* find(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = by_name.find(name);
        if (it == by_name.end()) {
            problems.push_back("missing " + name);
            return nullptr;
        }
        const mnx_weight_desc* d = it->second;
        bool ok = d->ndim == (int)shape.size() && d->data != nullptr;
        int i = 0;
        for (int64_t s : shape) ok = ok && d->shape[i++] == s;
        if (!ok) {
            std::string got = "[";
            for (int k = 0; k < d->ndim; ++k) got += std::to_string(d->shape[k]) + (k + 1 < d->ndim ? "," : "");
            problems.push_back("shape " + name + ": got " + got + "]");
            return nullptr;
        }
        return d;
    }
    void* dalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            problems.push_back("hipMalloc failed for " + std::to_string(bytes) + " bytes");
            return nullptr;
        }
        h->allocs.push_back(p);
        h->bytes += bytes;
        return p;
    }
    float* up32(const float* host, size_t n) {
        float* p = (float*)dalloc(n * sizeof(float));
        if (p && hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            problems.push_back("hipMemcpy H2D failed");
        return p;
    }
    float* f32(const std::string& name, std::initializer_list<int64_t> shape) {
        const mnx_weight_desc* d = find(name, shape);
        if (!d) return nullptr;
        size_t n = 1;
        for (int64_t s : shape) n *= (size_t)s;
        return up32(d->data, n);
    }
    W16 w16(const std::string& name, std::initializer_list<int64_t> shape) {
        W16 w;
        const mnx_weight_desc* d = find(name, shape);
        if (!d) return w;
        size_t n = 1;
        for (int64_t s : shape) n *= (size_t)s;
        if (n > staging_elems) {
            problems.push_back("staging too small for " + name);
            return w;
        }
        const int dt = h->dt;
        float scale = 1.f;
        if (dt == MNX_DT_F16X3) {
            // fp16 split: store 2^k W with max |2^k W| in [2^14, 2^15) — the lo plane of a weight of typical size
            // (|w| ~ 0.02) would be subnormal otherwise; the GEMM epilogue multiplies by 2^-k (exact)
            float mx = 0.f;
            for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(d->data[i]));
            if (mx > 0.f && std::isfinite(mx)) {
                int e = 0;
                std::frexp(mx, &e);                        // mx = f * 2^e, f in [0.5, 1)
                const int k = std::min(60, std::max(-60, 15 - e));
                scale = std::ldexp(1.f, k);
            }
        }
        w.p = dalloc(n * dt_size(dt));
        if (!w.p) return w;
        w.lo = dt_split(dt) ? n : 0;
        w.oscale = 1.f / scale;
        if (hipMemcpy(staging, d->data, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            launch_cast16(dt, staging, w.p, n, 0, w.lo, scale) != hipSuccess ||
            hipStreamSynchronize(0) != hipSuccess)
            problems.push_back("convert failed for " + name);
        return w;
    }
    const float* host(const std::string& name, std::initializer_list<int64_t> shape) {
        const mnx_weight_desc* d = find(name, shape);
        return d ? d->data : nullptr;
    }
};

int check_cfg(const mnx_config& c, std::string& why) {
    auto bad = [&](const char* m) { why = m; return MNX_ERR_INVALID_ARG; };
    if (c.n_stages < 1 || c.n_stages > 4) return bad("n_stages must be 1..4");
    if (c.patch != 4) return bad("patch must be 4");
    if (c.window != 12) return bad("window must be 12");
    if (c.embed_dim % 32 || c.embed_dim > 128) return bad("embed_dim must be 32..128, multiple of 32");
    int g = c.img_size / c.patch;
    if (c.img_size % c.patch) return bad("img_size not a multiple of patch");
    for (int s = 0; s < c.n_stages; ++s) {
        int C = c.embed_dim << s;
        if (c.heads[s] * 32 != C) return bad("head_dim must be 32 in every stage");
        if (g % c.window) return bad("every stage's token grid must be a multiple of the window (no padding path)");
        if (c.depths[s] < 1) return bad("depth < 1");
        if (s + 1 < c.n_stages) {
            if (g & 1) return bad("odd grid before merge");
            g /= 2;
        }
    }
    // g x g is the memory the decoder attends over: dec_attn_kernel scores two keys per thread (<= 512), the fused ticks hold
    // PS_CROSS = 160 (tick_tile falls back to the eight-launch tick above that); the reference's 384 x 384 gives 144
    if (g * g > 512) return bad("the encoder's last grid must have <= 512 positions (the cross-attention kernels hold 512 keys)");
    if (c.dec_dim != 256 || c.dec_heads != 8) return bad("decoder kernels are built for d_model 256, 8 heads");
    if (c.dec_layers < 1 || c.dec_layers > MAX_DEC_LAYERS) return bad("dec_layers out of range");
    if (c.dec_ff % 256 || c.dec_ff < 256) return bad("dec_ff must be a multiple of 256");
    if (c.vocab > 256 || c.vocab != c.sym_offset + 2 * c.coord_bins) return bad("vocab must be sym_offset+2*bins <= 256");
    if (c.max_len < 1 || c.max_len > 512) return bad("max_len must be 1..512");
    if (c.max_batch < 1) return bad("max_batch < 1");
    if (c.max_atoms < 1 || c.max_atoms > 256) return bad("max_atoms must be 1..256");
    if (c.compute_dtype < MNX_DTYPE_BF16 || c.compute_dtype > MNX_DTYPE_FP16X3M) return bad("compute_dtype");
    if (c.dec_slots < 0 || c.dec_slots > MAX_SLOTS || (c.dec_slots % ROW_TILE) != 0) return bad("dec_slots");
    if (c.pe_len < ROW_TILE) return bad("pe_len too small");
    return MNX_OK;
}

}  // namespace

extern "C" {

int mnx_abi_version(void) { return MNX_ABI_VERSION; }

const char* mnx_last_error(const mnx_engine* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

size_t mnx_workspace_bytes(const mnx_engine* h) { return h ? h->bytes : 0; }

void mnx_destroy(mnx_engine* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (const char* sp = getenv("MNX_FUSED_STAMPS")) { (void)hipDeviceSynchronize(); dec_fused_dump_stamps(sp); }   // lab aid
    for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    if (h->enc_stream) hipStreamDestroy(h->enc_stream);
    if (h->ev_order) hipEventDestroy(h->ev_order);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (int i = 1; i < MAX_TICK_BRANCHES; ++i) {
        if (h->tick_streams[i]) hipStreamDestroy(h->tick_streams[i]);
        if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]);
    }
    for (int i = 0; i < 2; ++i) {
        if (h->ev_enc_done[i]) hipEventDestroy(h->ev_enc_done[i]);
        if (h->ev_feat_free[i]) hipEventDestroy(h->ev_feat_free[i]);
        if (h->ev_poll[i]) hipEventDestroy(h->ev_poll[i]);
    }
    for (auto& ev : h->ev_pool) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    for (void* p : h->allocs) hipFree(p);
    if (h->host_flag) hipHostFree(h->host_flag);
    delete h;
}

int mnx_create(const mnx_config* cfg, const mnx_weight_desc* weights, int32_t n_weights, int32_t device,
               mnx_engine** out) {
    if (out) *out = nullptr;
    if (!cfg || !weights || !out || n_weights <= 0) {
        g_create_error = "mnx_create: null argument";
        return MNX_ERR_INVALID_ARG;
    }
    std::string why;
    if (check_cfg(*cfg, why) != MNX_OK) {
        g_create_error = "mnx_create: bad config: " + why;
        return MNX_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "mnx_create: no HIP device " + std::to_string(device) + " (libmolnextr_hip needs an MI355X; there is no CPU fallback)";
        return MNX_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        g_create_error = "mnx_create: cannot select device";
        return MNX_ERR_NO_DEVICE;
    }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_error = std::string("mnx_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return MNX_ERR_NO_DEVICE;
    }
    mnx_engine* h = new mnx_engine();
    h->cfg = *cfg;
    h->device = device;
    // FP16X3M = the FP16X3 kernels and weights with the op classes of MNX_FP16X3M_TWO_TERM on two product terms
    h->dt = cfg->compute_dtype == MNX_DTYPE_FP16X3M ? MNX_DT_F16X3 : cfg->compute_dtype;
    if (cfg->compute_dtype == MNX_DTYPE_FP16X3M) {
        const int by_stage[4] = MNX_FP16X3M_TWO_TERM_BY_STAGE;
        // the table is written for Swin-B's four stages; a shallower encoder (the tests' tiny one) keeps its LAST stages' rows
        const int first[4] = MNX_FP16X3M_FIRST_BLOCK_BY_STAGE;
        for (int st = 0; st < cfg->n_stages; ++st) {
            h->two_mask[st] = by_stage[st + 4 - cfg->n_stages];
            h->two_first[st] = first[st + 4 - cfg->n_stages];
        }
    }
    const char* ng = getenv("MNX_NO_GRAPH");
    h->use_graph = !(ng && ng[0] == '1');
    // MNX_ENC_CUS=n (default 256): the encoder's persistent kernels (gemm256x3_kernel: one 150 KB-LDS workgroup per CU for the
    // length of a launch; window_attn_pipe_kernel: two) are launched on n workgroups (2 n), so that 256 - n CUs stay free for
    // the decode stream's kernels while they run (DESIGN.md "co-residency": measured, not a win). The count is process-wide
    // (gemm256.hip keeps it) and is set by EVERY mnx_create: an engine created without the variable restores 256, so that no
    // engine inherits another one's value.
    {
        int n = 256;
        if (const char* e = getenv("MNX_ENC_CUS")) {
            n = atoi(e);
            if (n < 64 || n > 256) {
                g_create_error = "mnx_create: MNX_ENC_CUS must be 64..256";
                delete h;
                return MNX_ERR_INVALID_ARG;
            }
        }
        set_persistent_cus(n);
    }
    if (const char* e = getenv("MNX_DEC_BRANCH_ROWS")) h->dec_branch_rows = atoi(e);
    if (const char* e = getenv("MNX_DEC_BRANCH_MAX")) h->dec_branch_max = std::max(1, std::min(MAX_TICK_BRANCHES, atoi(e)));
    if (const char* e = getenv("MNX_DEC_XCD")) h->dec_xcd = atoi(e) != 0;
    if (const char* e = getenv("MNX_DEC_TILE")) h->dec_tile = atoi(e);              // 0: never use the fused tick
    if (const char* e = getenv("MNX_DEC_TILE_FF")) h->dec_tile_ff = atoi(e);
    if (const char* e = getenv("MNX_DEC_FUSED_MAX")) h->dec_fused_max = atoi(e);    // largest capacity that runs fused
    if (const char* e = getenv("MNX_DEC_MID_MAX")) h->dec_mid_max = atoi(e);        // largest capacity that runs the mid form
    // dec_fused.hip instantiates (attention rows, feed-forward rows) = (2, 4) (2, 8) (4, 4) (4, 8) (4, 16): 16-row feed-forward
    // tiles only go with 4-row attention tiles (auto picks 2 rows up to 64 rows of capacity) — refused here, not at the first tick
    if ((h->dec_tile != -1 && h->dec_tile != 0 && h->dec_tile != 2 && h->dec_tile != 4) ||
        (h->dec_tile_ff != 4 && h->dec_tile_ff != 8 && h->dec_tile_ff != 16) ||
        (h->dec_tile_ff == 16 && h->dec_tile != 4 && h->dec_tile != 0)) {
        g_create_error = "mnx_create: MNX_DEC_TILE must be -1 (auto), 0, 2 or 4 and MNX_DEC_TILE_FF 4, 8 or 16 (16 only with MNX_DEC_TILE=4)";
        delete h;
        return MNX_ERR_INVALID_ARG;
    }
    Packer P;
    P.h = h;
    for (int i = 0; i < n_weights; ++i)
        if (weights[i].name) P.by_name[weights[i].name] = &weights[i];
    const mnx_config& c = h->cfg;

    // staging buffer for fp32 -> 16-bit conversion: the largest GEMM weight
    size_t max_w = 0;
    for (int s = 0; s < c.n_stages; ++s) {
        size_t C = (size_t)c.embed_dim << s;
        max_w = std::max(max_w, 4 * C * C);
        if (s + 1 < c.n_stages) max_w = std::max(max_w, 8 * C * C);
    }
    P.staging_elems = max_w;
    if (hipMalloc((void**)&P.staging, max_w * sizeof(float)) != hipSuccess) {
        g_create_error = "mnx_create: hipMalloc(staging) failed";
        delete h;
        return MNX_ERR_HIP;
    }

    // ---------------- encoder weights ----------------
    const std::string T = "transformer.";
    {
        const int C = c.embed_dim;
        const float* pw = P.host(T + "patch_embed.proj.weight", {C, 3, 4, 4});
        if (pw) {  // [C][48] -> [48][C]
            std::vector<float> wt((size_t)48 * C);
            for (int ch = 0; ch < C; ++ch)
                for (int i = 0; i < 48; ++i) wt[(size_t)i * C + ch] = pw[(size_t)ch * 48 + i];
            h->pe_wt = P.up32(wt.data(), wt.size());
        }
        h->pe_b = P.f32(T + "patch_embed.proj.bias", {C});
        h->pe_g = P.f32(T + "patch_embed.norm.weight", {C});
        h->pe_beta = P.f32(T + "patch_embed.norm.bias", {C});
    }
    const int64_t NT = (2 * c.window - 1) * (2 * c.window - 1);
    h->stages.resize(c.n_stages);
    for (int s = 0; s < c.n_stages; ++s) {
        StageW& st = h->stages[s];
        const int64_t C = (int64_t)c.embed_dim << s;
        st.C = (int)C;
        st.heads = c.heads[s];
        for (int b = 0; b < c.depths[s]; ++b) {
            const std::string p = T + "layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            BlockW w{};
            w.ln1_g = P.f32(p + "norm1.weight", {C});
            w.ln1_b = P.f32(p + "norm1.bias", {C});
            w.table = P.f32(p + "attn.relative_position_bias_table", {NT, st.heads});
            w.qkv_w = P.w16(p + "attn.qkv.weight", {3 * C, C});
            w.qkv_b = P.f32(p + "attn.qkv.bias", {3 * C});
            w.proj_w = P.w16(p + "attn.proj.weight", {C, C});
            w.proj_b = P.f32(p + "attn.proj.bias", {C});
            w.ln2_g = P.f32(p + "norm2.weight", {C});
            w.ln2_b = P.f32(p + "norm2.bias", {C});
            w.fc1_w = P.w16(p + "mlp.fc1.weight", {4 * C, C});
            w.fc1_b = P.f32(p + "mlp.fc1.bias", {4 * C});
            w.fc2_w = P.w16(p + "mlp.fc2.weight", {C, 4 * C});
            w.fc2_b = P.f32(p + "mlp.fc2.bias", {C});
            st.blocks.push_back(w);
        }
        if (s + 1 < c.n_stages) {
            const std::string p = T + "layers." + std::to_string(s) + ".downsample.";
            st.m_g = P.f32(p + "norm.weight", {4 * C});
            st.m_b = P.f32(p + "norm.bias", {4 * C});
            st.m_w = P.w16(p + "reduction.weight", {2 * C, 4 * C});
        }
    }
    const int64_t CF = (int64_t)c.embed_dim << (c.n_stages - 1);
    h->fn_g = P.f32(T + "norm.weight", {CF});
    h->fn_b = P.f32(T + "norm.bias", {CF});

    // ---------------- decoder weights ----------------
    DecWeights& dw = h->dw;
    const int64_t D = c.dec_dim, FF = c.dec_ff, V = c.vocab;
    const int64_t S = (int64_t)(c.img_size / c.patch >> (c.n_stages - 1)) * (c.img_size / c.patch >> (c.n_stages - 1));
    dw.layers = c.dec_layers; dw.heads = c.dec_heads; dw.dff = c.dec_ff; dw.vocab = c.vocab; dw.vpad = (c.vocab + 7) & ~7;
    dw.sym_offset = c.sym_offset; dw.bins = c.coord_bins; dw.pe_len = c.pe_len; dw.enc_dim = (int)CF;
    const std::string Dp = "decoder.chartok_coords.";
    dw.w_enc = P.f32(Dp + "enc_trans_layer.0.weight", {D, CF});
    dw.b_enc = P.f32(Dp + "enc_trans_layer.0.bias", {D});
    dw.lnF_g = P.f32(Dp + "decoder.layer_norm.weight", {D});
    dw.lnF_b = P.f32(Dp + "decoder.layer_norm.bias", {D});
    dw.emb = P.f32(Dp + "embeddings.make_embedding.emb_luts.0.weight", {V, D});
    dw.bout = P.f32(Dp + "output_layer.bias", {V});
    {
        const float* wo = P.host(Dp + "output_layer.weight", {V, D});
        if (wo) {
            std::vector<float> t((size_t)D * dw.vpad, 0.f);
            for (int64_t v = 0; v < V; ++v)
                for (int64_t k = 0; k < D; ++k) t[(size_t)k * dw.vpad + v] = wo[v * D + k];
            dw.wout_t = P.up32(t.data(), t.size());
        }
        // sinusoid table: recomputed exactly as the reference builds it (MolNexTR/models/embedding.py:30-35);
        // if the checkpoint carries pe.pe it must agree.
        std::vector<float> pe((size_t)c.pe_len * D);
        for (int64_t pos = 0; pos < c.pe_len; ++pos)
            for (int64_t i = 0; i < D; i += 2) {
                const float div = expf((float)i * (float)(-(std::log(10000.0) / (double)D)));
                pe[pos * D + i] = sinf((float)pos * div);
                pe[pos * D + i + 1] = cosf((float)pos * div);
            }
        auto it = P.by_name.find(Dp + "embeddings.make_embedding.pe.pe");
        if (it != P.by_name.end() && it->second->data) {
            const mnx_weight_desc* d = it->second;
            if (d->ndim == 3 && d->shape[0] == c.pe_len && d->shape[1] == 1 && d->shape[2] == D)
                memcpy(pe.data(), d->data, pe.size() * sizeof(float));   // take the checkpoint's buffer verbatim
            else
                P.problems.push_back("shape " + Dp + "embeddings.make_embedding.pe.pe");
        }
        dw.pe = P.up32(pe.data(), pe.size());
    }
    // [rows][cols] -> device [cols][rows]: the fused tick reads a weight COLUMN per lane (dec_fused.hip)
    auto upT = [&](const float* src, int64_t rows, int64_t cols) -> const float* {
        if (!src) return nullptr;
        std::vector<float> t((size_t)rows * cols);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t k = 0; k < cols; ++k) t[(size_t)k * rows + r] = src[(size_t)r * cols + k];
        return P.up32(t.data(), t.size());
    };
    std::vector<float> memkv_w((size_t)c.dec_layers * 2 * D * D), memkv_b((size_t)c.dec_layers * 2 * D);
    bool memkv_ok = true;
    for (int l = 0; l < c.dec_layers; ++l) {
        const std::string p = Dp + "decoder.transformer_layers." + std::to_string(l) + ".";
        DecLayerW& L = dw.L[l];
        L.ln1_g = P.f32(p + "layer_norm_1.weight", {D});
        L.ln1_b = P.f32(p + "layer_norm_1.bias", {D});
        L.ln2_g = P.f32(p + "layer_norm_2.weight", {D});
        L.ln2_b = P.f32(p + "layer_norm_2.bias", {D});
        L.lnf_g = P.f32(p + "feed_forward.layer_norm.weight", {D});
        L.lnf_b = P.f32(p + "feed_forward.layer_norm.bias", {D});
        const float *wq = P.host(p + "self_attn.linear_query.weight", {D, D}), *bq = P.host(p + "self_attn.linear_query.bias", {D});
        const float *wk = P.host(p + "self_attn.linear_keys.weight", {D, D}), *bk = P.host(p + "self_attn.linear_keys.bias", {D});
        const float *wv = P.host(p + "self_attn.linear_values.weight", {D, D}), *bv = P.host(p + "self_attn.linear_values.bias", {D});
        if (wq && wk && wv && bq && bk && bv) {
            std::vector<float> w((size_t)3 * D * D), b((size_t)3 * D);
            memcpy(&w[0], wq, D * D * 4); memcpy(&w[D * D], wk, D * D * 4); memcpy(&w[2 * D * D], wv, D * D * 4);
            memcpy(&b[0], bq, D * 4); memcpy(&b[D], bk, D * 4); memcpy(&b[2 * D], bv, D * 4);
            L.wqkv = P.up32(w.data(), w.size());
            L.bqkv = P.up32(b.data(), b.size());
            L.wqkv_t = upT(w.data(), 3 * D, D);
        }
        L.wo = P.f32(p + "self_attn.final_linear.weight", {D, D});
        L.wo_t = upT(P.host(p + "self_attn.final_linear.weight", {D, D}), D, D);
        L.bo = P.f32(p + "self_attn.final_linear.bias", {D});
        L.wq2 = P.f32(p + "context_attn.linear_query.weight", {D, D});
        L.wq2_t = upT(P.host(p + "context_attn.linear_query.weight", {D, D}), D, D);
        L.bq2 = P.f32(p + "context_attn.linear_query.bias", {D});
        L.wo2 = P.f32(p + "context_attn.final_linear.weight", {D, D});
        L.wo2_t = upT(P.host(p + "context_attn.final_linear.weight", {D, D}), D, D);
        L.bo2 = P.f32(p + "context_attn.final_linear.bias", {D});
        const float *ck = P.host(p + "context_attn.linear_keys.weight", {D, D}), *cbk = P.host(p + "context_attn.linear_keys.bias", {D});
        const float *cv = P.host(p + "context_attn.linear_values.weight", {D, D}), *cbv = P.host(p + "context_attn.linear_values.bias", {D});
        if (ck && cv && cbk && cbv) {
            memcpy(&memkv_w[(size_t)l * 2 * D * D], ck, D * D * 4);
            memcpy(&memkv_w[(size_t)l * 2 * D * D + D * D], cv, D * D * 4);
            memcpy(&memkv_b[(size_t)l * 2 * D], cbk, D * 4);
            memcpy(&memkv_b[(size_t)l * 2 * D + D], cbv, D * 4);
        } else {
            memkv_ok = false;
        }
        L.w1 = P.f32(p + "feed_forward.w_1.weight", {FF, D});
        L.w1_t = upT(P.host(p + "feed_forward.w_1.weight", {FF, D}), FF, D);
        L.b1 = P.f32(p + "feed_forward.w_1.bias", {FF});
        L.w2 = P.f32(p + "feed_forward.w_2.weight", {D, FF});
        L.w2_t = upT(P.host(p + "feed_forward.w_2.weight", {D, FF}), D, FF);
        L.b2 = P.f32(p + "feed_forward.w_2.bias", {D});
    }
    if (memkv_ok) {
        dw.w_memkv = P.up32(memkv_w.data(), memkv_w.size());
        dw.b_memkv = P.up32(memkv_b.data(), memkv_b.size());
    }
    {
        const float *w1 = P.host("decoder.edges.mlp.0.weight", {D, 2 * D}), *b1 = P.host("decoder.edges.mlp.0.bias", {D});
        if (w1 && b1) {
            std::vector<float> w((size_t)2 * D * D), b((size_t)2 * D, 0.f);
            for (int64_t n = 0; n < D; ++n) {
                memcpy(&w[(size_t)n * D], w1 + n * 2 * D, D * 4);              // multiplies h_i
                memcpy(&w[(size_t)(D + n) * D], w1 + n * 2 * D + D, D * 4);    // multiplies h_j
                b[D + n] = b1[n];
            }
            dw.edge_w1cat = P.up32(w.data(), w.size());
            dw.edge_b1cat = P.up32(b.data(), b.size());
        }
        dw.edge_w2 = P.f32("decoder.edges.mlp.2.weight", {7, D});
        dw.edge_b2 = P.f32("decoder.edges.mlp.2.bias", {7});
    }
    hipFree(P.staging);

    // ---------------- workspace ----------------
    const size_t MB = (size_t)c.max_batch;
    const size_t G = c.img_size / c.patch, L0 = G * G, C0 = c.embed_dim;
    size_t max_qkv = 0, max_h = 0, max_xn = 0;
    {
        size_t Ls = L0, Cs = C0;
        for (int s = 0; s < c.n_stages; ++s) {
            max_qkv = std::max(max_qkv, Ls * 3 * Cs);
            max_h = std::max(max_h, Ls * 4 * Cs);
            max_xn = std::max(max_xn, Ls * Cs);
            if (s + 1 < c.n_stages) { Ls /= 4; Cs *= 2; }
        }
    }
    h->xa = (float*)P.dalloc(MB * L0 * C0 * 4);
    h->xb = (float*)P.dalloc(MB * L0 * C0 * 4 / 2);
    // operand bytes per element: 2 (bf16 / fp16), 4 (fp32 parity mode, or the two 16-bit planes of the split modes)
    const size_t es = dt_size(h->dt);
    // split modes: the lo plane follows the hi plane after PLANE_SKEW extra elements, so that the two planes of a row
    // are not a large power of two apart (same HBM channel / bank for every hi / lo pair of a stream)
    const size_t skew = dt_split(h->dt) ? PLANE_SKEW : 0;
    h->xn16 = P.dalloc(MB * max_xn * es + skew * 2);
    h->qkv16 = P.dalloc(MB * max_qkv * es + skew * 2);
    h->attn16 = P.dalloc(MB * max_xn * es + skew * 2);
    h->h16 = P.dalloc(MB * max_h * es + skew * 2);
    if (dt_split(h->dt)) {
        h->xn_lo = MB * max_xn + skew; h->qkv_lo = MB * max_qkv + skew; h->attn_lo = MB * max_xn + skew; h->h_lo = MB * max_h + skew;
    }
    {   // the reference's PatchMerging.reduction has no bias; every GEMM kernel adds this vector instead, so that the rows
        // of a layer that different kernels compute (launch_gemm16 splits by batch size) go through the same additions
        const size_t nz = std::max((size_t)c.embed_dim << c.n_stages, (size_t)4096);   // also the stand-in bias of mnx_gemm16_split
        h->zero_bias = (float*)P.dalloc(nz * sizeof(float));
        h->zero_bias_n = (int)nz;
        if (h->zero_bias && hipMemset(h->zero_bias, 0, nz * sizeof(float)) != hipSuccess) P.problems.push_back("hipMemset failed");
    }
    h->enc_flag = (int*)P.dalloc(sizeof(int));
    if (h->enc_flag && hipMemset(h->enc_flag, 0, sizeof(int)) != hipSuccess) P.problems.push_back("hipMemset failed");
    DecBuffers& db = h->db;
    const int SL = c.dec_slots > 0 ? c.dec_slots : 2048;
    h->n_chunk_bufs = SL / ROW_TILE;   // one reference batch per 32-slot row tile
    db.T = c.max_len; db.S = (int)S; db.slots = SL; db.mem_blocks = h->n_chunk_bufs * ROW_TILE; db.kmax = c.max_atoms;
    db.st = (DecState*)P.dalloc(sizeof(DecState));
    db.x = (float*)P.dalloc((size_t)SL * D * 4);
    db.x2 = (float*)P.dalloc((size_t)SL * D * 4);
    db.part = (float*)P.dalloc((size_t)(FF / 256) * SL * D * 4);
    // partial planes of the fused / mid tick: only ticks of up to max(dec_fused_max, dec_mid_max) rows of capacity touch them
    // (128 rows by default: 4 MB; every slot would be 100 MB at the bench's 3072)
    // (tick branches — off by default — run fused tiles at any row offset of a larger tick: every slot then)
    db.fpart_rows = h->dec_branch_rows > 0 ? SL
                  : std::min(SL, std::max(ROW_TILE, (std::max(h->dec_fused_max, h->dec_mid_max) + ROW_TILE - 1) / ROW_TILE * ROW_TILE));
    db.fpart = (float*)P.dalloc((size_t)2 * 16 * db.fpart_rows * D * 4);
    if (db.fpart && hipMemset(db.fpart, 0, (size_t)2 * 16 * db.fpart_rows * D * 4) != hipSuccess) P.problems.push_back("hipMemset failed");
    if (dec_fused_init() != hipSuccess) P.problems.push_back("dec_fused_init: LDS opt-in failed");
    db.q = (float*)P.dalloc((size_t)SL * D * 4);
    db.ctx = (float*)P.dalloc((size_t)SL * D * 4);
    db.h = (float*)P.dalloc((size_t)SL * FF * 4);
    // self K / V and projected memory K / V: 24-bit block fixed point, 100 bytes per cached row of 32 channels (kvq.h)
    db.Tq = kvq_rows(c.max_len); db.Sq = kvq_rows((int)S);
    const size_t cache = (size_t)c.dec_layers * SL * c.dec_heads * kvq_block_bytes(db.Tq);
    db.self_k = (char*)P.dalloc(cache);
    db.self_v = (char*)P.dalloc(cache);
    // every scale of a block must be finite before its first key is written (a lane whose key index is clamped may fetch a row
    // beyond the written ones and multiplies a zero probability by its scale)
    if (db.self_k && db.self_v && (hipMemset(db.self_k, 0, cache) != hipSuccess || hipMemset(db.self_v, 0, cache) != hipSuccess))
        P.problems.push_back("hipMemset failed");
    db.memory = (float*)P.dalloc((size_t)ROW_TILE * S * D * 4);
    db.mem_kv32 = (float*)P.dalloc((size_t)ROW_TILE * S * c.dec_layers * 2 * D * 4);
    const size_t mem_bytes = (size_t)db.mem_blocks * c.dec_layers * 2 * c.dec_heads * kvq_block_bytes(db.Sq);
    db.mem_kv = (char*)P.dalloc(mem_bytes);
    if (db.mem_kv && hipMemset(db.mem_kv, 0, mem_bytes) != hipSuccess) P.problems.push_back("hipMemset failed");
    db.tokens = (int*)P.dalloc((size_t)SL * c.max_len * 4);
    db.logp = (float*)P.dalloc((size_t)SL * c.max_len * 4);
    db.hidden = (float*)P.dalloc((size_t)SL * c.max_len * D * 4);
    db.edge_g = (float*)P.dalloc((size_t)ROW_TILE * db.kmax * D * 4);
    db.edge_uv = (float*)P.dalloc((size_t)ROW_TILE * db.kmax * 2 * D * 4);
    db.edge_prob = (float*)P.dalloc((size_t)ROW_TILE * db.kmax * db.kmax * 8 * 4);
    h->out_trace = nullptr;   // allocated lazily on first traced decode (test aid)
    const size_t ring_rows = std::max<size_t>(ROW_TILE, MB);   // one encode group (max_batch images) per buffer
    h->feat_ring[0] = (float*)P.dalloc(ring_rows * S * CF * 4);
    h->feat_ring[1] = (float*)P.dalloc(ring_rows * S * CF * 4);
    h->slot_lists = (int*)P.dalloc((size_t)MAX_CHUNKS * ROW_TILE * 4);
    h->tc_dev = (TokenClasses*)P.dalloc(sizeof(TokenClasses));
    h->prep_bbox = (int*)P.dalloc(4 * sizeof(int));   // at create: mnx_preprocess may run beside another entry point
    {
        // Encoder and decoder run concurrently on separate streams; the encoder stream gets the high priority (its
        // large GEMM grids otherwise queue behind the decode ticks' many small kernels: measured +1.6 %). Partitioning
        // the chip with CU masks instead was measured and rejected.
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&h->enc_stream, hipStreamNonBlocking, hi) != hipSuccess) P.problems.push_back("stream create failed");
        if (hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming) != hipSuccess) P.problems.push_back("event create failed");
        if (h->dec_branch_rows > 0) {       // tick branches (off by default): their streams / events only when asked for
            if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) P.problems.push_back("event create failed");
            for (int i = 1; i < MAX_TICK_BRANCHES; ++i)
                if (hipStreamCreateWithFlags(&h->tick_streams[i], hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess)
                    P.problems.push_back("tick branch stream / event create failed");
        }
    }
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&h->ev_enc_done[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_feat_free[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_poll[i], hipEventDisableTiming) != hipSuccess)
            P.problems.push_back("event create failed");
    if (hipHostMalloc((void**)&h->host_flag, 32768) != hipSuccess) P.problems.push_back("hipHostMalloc failed");

    if (!P.problems.empty()) {
        g_create_error = "mnx_create: " + std::to_string(P.problems.size()) + " problem(s):";
        for (size_t i = 0; i < P.problems.size() && i < 12; ++i) g_create_error += "\n  " + P.problems[i];
        bool weights_bad = false;
        for (auto& p : P.problems) weights_bad = weights_bad || p.rfind("missing", 0) == 0 || p.rfind("shape", 0) == 0;
        mnx_destroy(h);
        return weights_bad ? MNX_ERR_WEIGHTS : MNX_ERR_HIP;
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        g_create_error = "mnx_create: device sync failed";
        mnx_destroy(h);
        return MNX_ERR_HIP;
    }
    *out = h;
    return MNX_OK;
}

int mnx_set_encoder_tap(mnx_engine* h, int32_t item, float* dst) {
    if (!h) return MNX_ERR_INVALID_ARG;
    h->tap_item = item;
    h->tap_dst = dst;
    return MNX_OK;
}

int mnx_set_split_terms(mnx_engine* h, int32_t mask) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (mask < 0 || mask > SPL_ALL) { h->err = "mnx_set_split_terms: mask must be 0..63"; return MNX_ERR_INVALID_ARG; }
    h->split_mask = mask;
    return MNX_OK;
}

int mnx_set_op_terms(mnx_engine* h, int32_t stage, int32_t two_term_mask, int32_t first_block, int32_t last_block) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (h->dt != MNX_DT_F16X3) { h->err = "mnx_set_op_terms: compute_dtype must be FP16X3 or FP16X3M"; return MNX_ERR_INVALID_ARG; }
    if (stage < -1 || stage >= h->cfg.n_stages) { h->err = "mnx_set_op_terms: stage must be -1 (all) or 0..n_stages-1"; return MNX_ERR_INVALID_ARG; }
    if (two_term_mask < 0 || two_term_mask > SPL_ALL || (two_term_mask & SPL_ATTN)) {
        h->err = "mnx_set_op_terms: mask must be a subset of the Linear classes (1 qkv, 4 proj, 8 fc1, 16 fc2, 32 merge)";
        return MNX_ERR_INVALID_ARG;
    }
    if (first_block < 0 || last_block < first_block) { h->err = "mnx_set_op_terms: 0 <= first_block <= last_block required"; return MNX_ERR_INVALID_ARG; }
    for (int st = 0; st < h->cfg.n_stages; ++st)
        if (stage < 0 || stage == st) { h->two_mask[st] = two_term_mask; h->two_first[st] = first_block; h->two_last[st] = last_block; }
    return MNX_OK;
}

int mnx_encoder_status(mnx_engine* h, int32_t* nonfinite, void* stream) {
    if (!h || !nonfinite) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->enc_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (flag) HIPCHK(h, hipMemsetAsync(h->enc_flag, 0, sizeof(int), s));
    *nonfinite = flag;
    return MNX_OK;
}

// mnx_predict / mnx_predict_beam: after the final synchronisation, turn a non-finite encoder output into an error
static int check_encoder_range(mnx_engine* h, hipStream_t s) {
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, h->enc_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (!flag) return MNX_OK;
    HIPCHK(h, hipMemsetAsync(h->enc_flag, 0, sizeof(int), s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->err = "encoder features are not finite (an activation left the fp16 range of compute_dtype FP16 / FP16X3, or the "
             "input holds NaN / Inf): use MNX_DTYPE_BF16X3 or MNX_DTYPE_FP32 for this checkpoint";
    return MNX_ERR_RANGE;
}

int mnx_encode(mnx_engine* h, const float* images, int32_t B, float* features_out, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!images || !features_out || B < 1) { h->err = "mnx_encode: null/empty argument"; return MNX_ERR_INVALID_ARG; }
    if (B > h->cfg.max_batch) { h->err = "mnx_encode: B exceeds max_batch"; return MNX_ERR_CAPACITY; }
    hipStream_t s = (hipStream_t)stream;
    const mnx_config& c = h->cfg;
    const int dt = h->dt;
    HIPCHK(h, hipSetDevice(h->device));
    int Hh = c.img_size / c.patch, Ww = Hh, C = c.embed_dim;
    float* cur = h->xa;
    float* other = h->xb;
    int item = 0;
    auto tap = [&](size_t elems) -> hipError_t {
        hipError_t e = hipSuccess;
        if (h->tap_item == item && h->tap_dst)
            e = hipMemcpyAsync(h->tap_dst, cur, elems * sizeof(float), hipMemcpyDeviceToDevice, s);
        ++item;
        return e;
    };
    // sampled measurement: every prof_stride-th encode call, at most 16 calls per enable
    const bool bracket = h->profiling && (h->prof_calls++ % h->prof_stride) == 0 && h->prof_groups < h->prof_max_groups;
    if (bracket) ++h->prof_groups;
    // brackets one launch with a pair of HIP events on the stream it is launched on (sampled encode calls only)
    auto timed = [&](int kind, double work, auto&& launch) -> hipError_t {
        if (!bracket) return launch();
        if (h->ev_used == h->ev_pool.size()) {
            mnx_engine::Ev ev{};
            hipError_t e1 = hipEventCreate(&ev.a), e2 = hipEventCreate(&ev.b);
            if (e1 != hipSuccess || e2 != hipSuccess) return e1 != hipSuccess ? e1 : e2;
            h->ev_pool.push_back(ev);
        }
        mnx_engine::Ev& ev = h->ev_pool[h->ev_used++];
        ev.work = work; ev.kind = kind;
        hipError_t e0 = hipEventRecord(ev.a, s);
        if (e0 != hipSuccess) return e0;
        e0 = launch();
        if (e0 != hipSuccess) return e0;
        return hipEventRecord(ev.b, s);
    };
    const double es = (double)dt_size(dt);
    const bool split = dt_split(dt);
    // split modes: a_lo / c_lo = lo-plane offsets of the activation buffers, cls = the op class whose bit of split_mask
    // selects three product terms (default) or the hi.hi term alone (error-budget aid)
    // terms of an op class: its full count (3, or 2 for the classes of two_mask) or hi.hi alone. A 16-bit activation is
    // written as ONE plane when its consumer runs on two terms (planes_for): half the bytes out of the producer, half into
    // the consumer; the hi plane is the same bits either way.
    int stage_now = 0, block_now = 0;
    auto terms_of = [&](int cls) {
        if (!(h->split_mask & cls)) return 1;
        const bool in_range = block_now >= h->two_first[stage_now] && block_now <= h->two_last[stage_now];
        return (in_range && (h->two_mask[stage_now] & cls)) ? 2 : 3;
    };
    auto planes_for = [&](int consumer_cls) { return split && terms_of(consumer_cls) == 2 ? 1 : 2; };
    auto gemm = [&](int epi, const void* A, size_t a_lo, const W16& Wt, void* Cc, size_t c_lo, const float* bias,
                    const float* resid, int M, int N, int K, int cls, int c_planes = 2) -> hipError_t {
        SplitArgs sp;
        sp.a_lo = a_lo; sp.w_lo = Wt.lo; sp.c_lo = c_lo; sp.oscale = Wt.oscale;
        sp.terms = terms_of(cls);
        sp.c_planes = c_planes;
        // kind 4: the Linear layers of the blocks with C >= 512 (Swin-B stages 3 and 4, the MFMA-bound shapes); kind 0: the rest
        return timed((cls != SPL_MERGE && std::min(N, K) >= 512) ? 4 : 0, 2.0 * (double)M * (double)N * (double)K,
                     [&]() { return launch_gemm16(dt, epi, A, Wt.p, Cc, bias, resid, M, N, K, s, split ? &sp : nullptr); });
    };
    auto ln = [&](const float* x, const float* g, const float* b, void* y16, float* y32, int M, int Cc, int planes = 2) -> hipError_t {
        return timed(1, (double)M * Cc * (4.0 + (y16 ? es * planes / 2 : 0.0) + (y32 ? 4.0 : 0.0)),
                     [&]() { return launch_layernorm16(dt, x, g, b, y16, y32, M, Cc, 1e-5f, s, h->xn_lo, y32 ? h->enc_flag : nullptr, planes); });
    };
    HIPCHK(h, timed(3, (double)B * (3.0 * c.img_size * c.img_size + (double)Hh * Ww * C) * 4.0,
                    [&]() { return launch_patch_embed(images, h->pe_wt, h->pe_b, h->pe_g, h->pe_beta, cur, B, c.img_size, C, s); }));
    HIPCHK(h, tap((size_t)B * Hh * Ww * C));
    for (int si = 0; si < c.n_stages; ++si) {
        StageW& st = h->stages[si];
        stage_now = si;
        const int M = B * Hh * Ww;
        for (size_t bi = 0; bi < st.blocks.size(); ++bi) {
            const BlockW& w = st.blocks[bi];
            block_now = (int)bi;
            const int shift = (bi % 2 == 0) ? 0 : c.window / 2;   // reference transformers.py:363
            HIPCHK(h, ln(cur, w.ln1_g, w.ln1_b, h->xn16, nullptr, M, C, planes_for(SPL_QKV)));
            HIPCHK(h, gemm(EPI_BIAS_16, h->xn16, h->xn_lo, w.qkv_w, h->qkv16, h->qkv_lo, w.qkv_b, nullptr, M, 3 * C, C, SPL_QKV));
            HIPCHK(h, timed(2, (double)M * C * 4.0 * es, [&]() {
                return launch_window_attn(dt, h->qkv16, w.table, h->attn16, B, Hh, Ww, C, st.heads, shift, s, h->qkv_lo,
                                          h->attn_lo, (h->split_mask & SPL_ATTN) ? 3 : 1);
            }));
            HIPCHK(h, gemm(EPI_RESID_F32, h->attn16, h->attn_lo, w.proj_w, cur, 0, w.proj_b, cur, M, C, C, SPL_PROJ));
            HIPCHK(h, ln(cur, w.ln2_g, w.ln2_b, h->xn16, nullptr, M, C, planes_for(SPL_FC1)));
            HIPCHK(h, gemm(EPI_GELU_16, h->xn16, h->xn_lo, w.fc1_w, h->h16, h->h_lo, w.fc1_b, nullptr, M, 4 * C, C, SPL_FC1, planes_for(SPL_FC2)));
            HIPCHK(h, gemm(EPI_RESID_F32, h->h16, h->h_lo, w.fc2_w, cur, 0, w.fc2_b, cur, M, C, 4 * C, SPL_FC2));
            HIPCHK(h, tap((size_t)M * C));
        }
        if (si + 1 < c.n_stages) {
            block_now = (int)st.blocks.size() - 1;        // the reduction behind the stage counts as its last block
            HIPCHK(h, launch_merge_ln16(dt, cur, st.m_g, st.m_b, h->xn16, B, Hh, Ww, C, 1e-5f, s, h->xn_lo, planes_for(SPL_MERGE)));
            HIPCHK(h, gemm(EPI_BIAS_F32, h->xn16, h->xn_lo, st.m_w, other, 0, h->zero_bias, nullptr, M / 4, 2 * C, 4 * C, SPL_MERGE));
            std::swap(cur, other);
            Hh /= 2; Ww /= 2; C *= 2;
            HIPCHK(h, tap((size_t)B * Hh * Ww * C));
        }
    }
    HIPCHK(h, ln(cur, h->fn_g, h->fn_b, nullptr, features_out, B * Hh * Ww, C));
    return MNX_OK;
}

// row tiles of the fused greedy tick for a capacity of `rows` rows: 100 x attention tile + feed-forward tile (0: the
// decoder.hip tick)
static int tick_tile(const mnx_engine* h, int rows) {
    const mnx_config& c = h->cfg;
    if (c.dec_ff != 1024 || c.dec_heads != 8 || c.dec_dim != 256 || c.max_len + 1 > 512 || h->db.S > 160) return 0;
    if (h->dec_tile == 0 || rows % 16 || rows > h->db.fpart_rows) return 0;
    if (rows > h->dec_fused_max) {      // mid form: 4-row attention tiles, 16-row feed-forward tiles
        if (rows > h->dec_mid_max) return 0;
        return 2000 + 100 * 4 + 16;
    }
    const int r = h->dec_tile > 0 ? h->dec_tile : (rows <= 64 ? 2 : 4);
    return (h->dec_xcd ? 1000 : 0) + 100 * r + h->dec_tile_ff;
}

// Branches of a tick of `rows` rows of capacity: (first row, rows) pairs, multiples of 32 rows, covering [0, rows).
static int tick_branches(const mnx_engine* h, int rows, bool single, int (*br)[2]) {
    int nb = 1;
    if (!single && h->dec_branch_rows > 0 && rows > h->dec_branch_rows)
        nb = std::min(h->dec_branch_max, (rows + h->dec_branch_rows - 1) / h->dec_branch_rows);
    const int tiles = rows / ROW_TILE;
    nb = std::max(1, std::min(nb, tiles));
    int base = 0;
    for (int i = 0; i < nb; ++i) {
        const int t = tiles / nb + (i < tiles % nb ? 1 : 0);
        br[i][0] = base; br[i][1] = t * ROW_TILE;
        base += t * ROW_TILE;
    }
    return nb;
}

// One tick on stream s: the begin kernel, then the layers + head of every branch — branch 0 on s, the others on the
// engine's branch streams between a fork event and join events (under stream capture this records parallel branches of the
// graph; without a graph it runs the same way eagerly).
static hipError_t enqueue_tick(mnx_engine* h, int slots, int rows, float* trace, int trace_rows, hipStream_t s,
                               const int* forced) {
    int br[MAX_TICK_BRANCHES][2];
    const int nb = tick_branches(h, rows, trace != nullptr || forced != nullptr, br);
    hipError_t e = dec_enqueue_status(h->db, slots, s);       // the begin kernel
    if (e != hipSuccess) return e;
    if (nb > 1) {
        if ((e = hipEventRecord(h->ev_fork, s)) != hipSuccess) return e;
        for (int i = 1; i < nb; ++i)
            if ((e = hipStreamWaitEvent(h->tick_streams[i], h->ev_fork, 0)) != hipSuccess) return e;
    }
    for (int i = 0; i < nb; ++i) {
        hipStream_t si = i == 0 ? s : h->tick_streams[i];
        e = dec_enqueue_tick_rows(h->dw, h->db, br[i][0], br[i][1], trace, trace_rows, si, nullptr, forced, tick_tile(h, br[i][1]));
        if (e != hipSuccess) return e;
    }
    for (int i = 1; i < nb; ++i) {
        if ((e = hipEventRecord(h->ev_join[i], h->tick_streams[i])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(s, h->ev_join[i], 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

static int get_tick_graph(mnx_engine* h, int slots, int rows, float* trace, int trace_rows, hipStream_t s,
                          hipGraphExec_t* out, const int* forced = nullptr) {
    *out = nullptr;
    if (!h->use_graph) return MNX_OK;
    int br[MAX_TICK_BRANCHES][2];
    const int nb = tick_branches(h, rows, trace != nullptr || forced != nullptr, br);
    GraphKey key{slots, rows, trace ? trace_rows : 0, forced ? trace_rows : 0, tick_tile(h, br[0][1]), nb};
    auto it = h->graphs.find(key);
    if (it != h->graphs.end()) { *out = it->second; return MNX_OK; }
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipError_t e = enqueue_tick(h, slots, rows, trace, trace_rows, s, forced);
    hipError_t e2 = hipStreamEndCapture(s, &g);
    if (e != hipSuccess || e2 != hipSuccess) {
        h->err = std::string("decode tick capture failed: ") + hipGetErrorString(e != hipSuccess ? e : e2);
        return MNX_ERR_HIP;
    }
    HIPCHK(h, hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    h->graphs[key] = exec;
    *out = exec;
    return MNX_OK;
}

static int run_ticks(mnx_engine* h, hipGraphExec_t exec, int slots, int rows, float* trace, int trace_rows, int n,
                     hipStream_t s, const int* forced = nullptr) {
    for (int i = 0; i < n; ++i) {
        if (exec) HIPCHK(h, hipGraphLaunch(exec, s));
        else HIPCHK(h, enqueue_tick(h, slots, rows, trace, trace_rows, s, forced));
    }
    return MNX_OK;
}

static int decode_greedy_impl(mnx_engine* h, const float* features, int32_t B, const int32_t* chunk_id, int32_t max_len,
                              int32_t stop_on_eos, const int32_t* forced_ids, int32_t* tokens, int32_t* lengths,
                              float* token_logp, float* hidden, float* logits_trace, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!features || !tokens || !lengths || B < 1) { h->err = "mnx_decode_greedy: null/empty argument"; return MNX_ERR_INVALID_ARG; }
    if (B > ROW_TILE || max_len < 1 || max_len > h->cfg.max_len) {
        h->err = "mnx_decode_greedy: B must be <= 32 and max_len <= cfg.max_len";
        return MNX_ERR_CAPACITY;
    }
    hipStream_t s = (hipStream_t)stream;
    const mnx_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(h->device));
    if (!s) {   // a default-flag stream synchronises implicitly with the null stream on both ends
        if (!h->own_stream) HIPCHK(h, hipStreamCreate(&h->own_stream));
        s = h->own_stream;
    }
    const int S = h->db.S, D = c.dec_dim;
    // enc_transform, then the cross-attention K/V of all layers in one SGEMM (memory block i = row i)
    HIPCHK(h, launch_sgemm_tn(features, h->dw.w_enc, h->dw.b_enc, h->db.memory, B * S, D, h->dw.enc_dim, s));
    HIPCHK(h, launch_sgemm_tn(h->db.memory, h->dw.w_memkv, h->dw.b_memkv, h->db.mem_kv32, B * S, c.dec_layers * 2 * D, D, s, S));
    HIPCHK(h, kvq_pack_enqueue(h->db.mem_kv32, h->db.mem_kv, B * c.dec_layers * 2 * c.dec_heads, S, h->db.Sq, s));
    HIPCHK(h, dec_enqueue_reset(h->db, s));
    HIPCHK(h, dec_enqueue_admit_rows(h->db, chunk_id, B, max_len, stop_on_eos, s));
    float* trace = nullptr;
    if (logits_trace) {
        if (!h->out_trace) {
            HIPCHK(h, hipMalloc((void**)&h->out_trace, (size_t)c.max_len * ROW_TILE * c.vocab * 4));
            h->allocs.push_back(h->out_trace);
            h->bytes += (size_t)c.max_len * ROW_TILE * c.vocab * 4;
        }
        trace = h->out_trace;
    }
    const int* forced = nullptr;
    if (forced_ids) {    // teacher forcing: the caller's [B, max_len] ids, re-strided to the state's [slot][T] rows
        if (!h->forced_ids) {
            const size_t bytes = (size_t)ROW_TILE * c.max_len * 4;
            HIPCHK(h, hipMalloc((void**)&h->forced_ids, bytes));
            h->allocs.push_back(h->forced_ids);
            h->bytes += bytes;
        }
        HIPCHK(h, hipMemcpy2DAsync(h->forced_ids, (size_t)h->db.T * 4, forced_ids, (size_t)max_len * 4, (size_t)max_len * 4, B,
                                   hipMemcpyDeviceToDevice, s));
        forced = h->forced_ids;
    }
    hipGraphExec_t exec = nullptr;
    int rc = get_tick_graph(h, ROW_TILE, ROW_TILE, trace, B, s, &exec, forced);
    if (rc != MNX_OK) return rc;
    const int poll = 8;
    for (int t = 0; t < max_len;) {
        const int n = std::min(poll, max_len - t);
        rc = run_ticks(h, exec, ROW_TILE, ROW_TILE, trace, B, n, s, forced);
        if (rc != MNX_OK) return rc;
        t += n;
        HIPCHK(h, dec_enqueue_status(h->db, ROW_TILE, s));
        HIPCHK(h, hipMemcpyAsync(h->host_flag, &h->db.st->n_active, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
        if (*h->host_flag == 0) break;
    }
    HIPCHK(h, gather_enqueue(h->db, nullptr, B, max_len, tokens, lengths, token_logp, hidden, s));
    if (logits_trace) HIPCHK(h, hipMemcpyAsync(logits_trace, trace, (size_t)max_len * B * c.vocab * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return MNX_OK;
}

int mnx_decode_greedy(mnx_engine* h, const float* features, int32_t B, const int32_t* chunk_id, int32_t max_len,
                      int32_t stop_on_eos, int32_t* tokens, int32_t* lengths, float* token_logp, float* hidden,
                      float* logits_trace, void* stream) {
    return decode_greedy_impl(h, features, B, chunk_id, max_len, stop_on_eos, nullptr, tokens, lengths, token_logp, hidden,
                              logits_trace, stream);
}

int mnx_decode_forced(mnx_engine* h, const float* features, int32_t B, const int32_t* chunk_id, int32_t max_len,
                      const int32_t* forced_ids, int32_t* argmax_ids, int32_t* lengths, float* forced_logp,
                      float* logits_trace, void* stream) {
    if (h && !forced_ids) { h->err = "mnx_decode_forced: forced_ids is null"; return MNX_ERR_INVALID_ARG; }
    return decode_greedy_impl(h, features, B, chunk_id, max_len, 1, forced_ids, argmax_ids, lengths, forced_logp, nullptr,
                              logits_trace, stream);
}

// Beam search over G reference batches in ONE step sequence: batch g = images [g ref_batch, (g + 1) ref_batch) of n_total
// (features at feats[g]), every image K hypotheses, one row per hypothesis — G x ref_batch x K rows per step. Images are
// independent; the positional-encoding rows are numbered inside each reference batch (beam_begin_kernel), so the result is
// that of G separate searches. Outputs [n_total, n_best, ...].
static int decode_beam_groups(mnx_engine* h, const float* const* feats, int G, int ref_batch, int n_total, int beam, int n_best,
                              int max_len, int32_t* tokens, int32_t* lengths, float* scores, float* hidden, hipStream_t s) {
    const mnx_config& c = h->cfg;
    const int S = h->db.S, D = c.dec_dim, T = h->db.T, B = n_total;
    BeamBuffers& bm = h->beam;
    auto lazy = [&](void** p, size_t bytes) -> hipError_t {
        if (*p) return hipSuccess;
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) { h->allocs.push_back(*p); h->bytes += bytes; }
        return e;
    };
    // capacities: MAX_BEAM_IMGS images x MAX_BEAM hypotheses of state; 256 kept hypotheses (32 images x 8 ... 256 x 1)
    constexpr int POOL = ROW_TILE * MAX_BEAM;
    const int pool_stride = std::min(MAX_BEAM, POOL / B);
    if (pool_stride < n_best) { h->err = "beam search: n_best x images exceeds the hypothesis pool (256)"; return MNX_ERR_CAPACITY; }
    HIPCHK(h, lazy((void**)&bm.bs, sizeof(BeamState)));
    HIPCHK(h, lazy((void**)&bm.blp, (size_t)MAX_BEAM_IMGS * MAX_BEAM * BEAM_LP_STRIDE * 4));
    HIPCHK(h, lazy((void**)&bm.anc, (size_t)MAX_BEAM_IMGS * MAX_BEAM * (T + 1) * 4));
    HIPCHK(h, lazy((void**)&bm.ptok, (size_t)POOL * T * 4));
    if (hidden) HIPCHK(h, lazy((void**)&bm.phid, (size_t)POOL * T * D * 4));
    bm.B = B; bm.K = beam; bm.n_best = n_best; bm.anc_stride = T + 1; bm.ref_batch = ref_batch; bm.pool_stride = pool_stride;
    BeamBuffers run = bm;
    if (!hidden) run.phid = nullptr;
    for (int g = 0; g < G; ++g) {      // enc_transform + memory K / V of batch g -> memory blocks g ref_batch ...
        const int n = std::min(ref_batch, B - g * ref_batch);
        char* memkv = h->db.mem_kv + (size_t)g * ref_batch * c.dec_layers * 2 * c.dec_heads * kvq_block_bytes(h->db.Sq);
        HIPCHK(h, launch_sgemm_tn(feats[g], h->dw.w_enc, h->dw.b_enc, h->db.memory, n * S, D, h->dw.enc_dim, s));
        HIPCHK(h, launch_sgemm_tn(h->db.memory, h->dw.w_memkv, h->dw.b_memkv, h->db.mem_kv32, n * S, c.dec_layers * 2 * D, D, s, S));
        HIPCHK(h, kvq_pack_enqueue(h->db.mem_kv32, memkv, n * c.dec_layers * 2 * c.dec_heads, S, h->db.Sq, s));
    }
    HIPCHK(h, dec_enqueue_reset(h->db, s));
    HIPCHK(h, beam_enqueue_init(h->db, run, max_len, s));
    const int rows = (B * beam + ROW_TILE - 1) / ROW_TILE * ROW_TILE;
    // one step = begin + 6 layers + head + pick, captured once per call (its arguments depend on B / beam / n_best)
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    if (h->use_graph) {
        HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipError_t e = dec_enqueue_tick(h->dw, h->db, rows, rows, nullptr, 0, s, &run);
        hipError_t e2 = hipStreamEndCapture(s, &g);
        if (e != hipSuccess || e2 != hipSuccess) {
            h->err = std::string("beam step capture failed: ") + hipGetErrorString(e != hipSuccess ? e : e2);
            return MNX_ERR_HIP;
        }
        HIPCHK(h, hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
    }
    struct ExecGuard {      // the per-call graph is released on every exit path
        hipGraphExec_t& e;
        ~ExecGuard() { if (e) { hipGraphExecDestroy(e); e = nullptr; } }
    } guard{exec};
    int rc = MNX_OK;
    const int poll = 8;
    for (int t = 0; t < max_len && rc == MNX_OK;) {
        const int n = std::min(poll, max_len - t);
        for (int i = 0; i < n; ++i) {
            hipError_t e = exec ? hipGraphLaunch(exec, s) : dec_enqueue_tick(h->dw, h->db, rows, rows, nullptr, 0, s, &run);
            if (e != hipSuccess) { h->err = std::string("beam step: ") + hipGetErrorString(e); rc = MNX_ERR_HIP; break; }
        }
        if (rc != MNX_OK) break;
        t += n;
        hipError_t e = dec_enqueue_status(h->db, rows, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h->host_flag, &h->db.st->n_active, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { h->err = std::string("beam poll: ") + hipGetErrorString(e); rc = MNX_ERR_HIP; break; }
        if (*h->host_flag == 0) break;
    }
    if (rc != MNX_OK) return rc;
    HIPCHK(h, beam_enqueue_gather(h->db, run, max_len, tokens, lengths, scores, hidden, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return MNX_OK;
}

int mnx_decode_beam(mnx_engine* h, const float* features, int32_t B, int32_t beam, int32_t n_best, int32_t max_len,
                    int32_t* tokens, int32_t* lengths, float* scores, float* hidden, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!features || !tokens || !lengths || !scores || B < 1) {
        h->err = "mnx_decode_beam: null/empty argument";
        return MNX_ERR_INVALID_ARG;
    }
    const mnx_config& c = h->cfg;
    if (B > ROW_TILE || beam < 1 || beam > MAX_BEAM || n_best < 1 || n_best > beam || max_len < 1 ||
        max_len > c.max_len || c.max_len + 1 > BEAM_ANC_MAX || c.vocab > BEAM_LP_STRIDE || B * beam > h->db.slots) {
        h->err = "mnx_decode_beam: B <= 32, 1 <= n_best <= beam <= 8, max_len <= cfg.max_len (<= 511), B x beam <= dec_slots required";
        return MNX_ERR_CAPACITY;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    if (!s) {
        if (!h->own_stream) HIPCHK(h, hipStreamCreate(&h->own_stream));
        s = h->own_stream;
    }
    return decode_beam_groups(h, &features, 1, B, B, beam, n_best, max_len, tokens, lengths, scores, hidden, s);
}

int mnx_predict_beam(mnx_engine* h, const float* images, int32_t n_img, int32_t ref_batch, int32_t beam, int32_t max_len,
                     int32_t* tokens, int32_t* lengths, float* scores, int32_t* n_atoms, int32_t* atom_idx,
                     uint8_t* edges, int32_t kmax, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!images || !tokens || !lengths || !scores || !n_atoms || !atom_idx || !edges || n_img < 1) {
        h->err = "mnx_predict_beam: null/empty argument";
        return MNX_ERR_INVALID_ARG;
    }
    if (!h->have_tc) { h->err = "mnx_predict_beam: call mnx_set_token_classes first"; return MNX_ERR_INVALID_ARG; }
    const mnx_config& c = h->cfg;
    if (ref_batch < 1 || ref_batch > ROW_TILE || ref_batch > c.max_batch || beam < 1 || beam > MAX_BEAM || max_len < 1 ||
        max_len > c.max_len || kmax < 1 || kmax > h->db.kmax || c.max_len + 1 > BEAM_ANC_MAX || c.vocab > BEAM_LP_STRIDE ||
        ref_batch * beam > h->db.slots) {
        h->err = "mnx_predict_beam: ref_batch <= min(32, max_batch), beam <= 8, max_len <= cfg.max_len (<= 511), kmax <= cfg.max_atoms, "
                 "ref_batch x beam <= dec_slots required";
        return MNX_ERR_CAPACITY;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    if (!s) {
        if (!h->own_stream) HIPCHK(h, hipStreamCreate(&h->own_stream));
        s = h->own_stream;
    }
    const int S = h->db.S, D = c.dec_dim;
    const size_t img_elems = (size_t)3 * c.img_size * c.img_size;
    // Reference batches searched together (one step sequence, decode_beam_groups): up to MNX_BEAM_GROUPS (default 8) batches
    // of one encoder launch group — a step of 4 x 160 rows costs 1.5x a step of 160 —, bounded by the state
    // capacity (MAX_BEAM_IMGS images, dec_slots rows) and by the memory blocks
    int g_max = 8;
    if (const char* e = getenv("MNX_BEAM_GROUPS")) g_max = std::max(1, atoi(e));
    g_max = std::max(1, std::min({g_max, MAX_BEAM_IMGS / ref_batch, h->db.slots / (ref_batch * beam), h->db.mem_blocks / ref_batch}));
    if (!h->beam_hidden) {      // decoder outputs along the winning hypotheses of the reference batches of one search
        const size_t bytes = (size_t)MAX_BEAM_IMGS * c.max_len * D * 4;
        HIPCHK(h, hipMalloc((void**)&h->beam_hidden, bytes));
        h->allocs.push_back(h->beam_hidden);
        h->bytes += bytes;
    }
    struct ExitGuard {          // every exit path: nothing of this call may still be in flight
        mnx_engine* h; hipStream_t s;
        ~ExitGuard() { (void)hipStreamSynchronize(h->enc_stream); (void)hipStreamSynchronize(s); }
    } exit_guard{h, s};
    // the encoder stream must not start before the caller's stream reaches this point (images ready)
    HIPCHK(h, hipMemsetAsync(h->enc_flag, 0, sizeof(int), s));     // the range flag is per call (see mnx_predict)
    HIPCHK(h, hipEventRecord(h->ev_poll[0], s));
    HIPCHK(h, hipStreamWaitEvent(h->enc_stream, h->ev_poll[0], 0));
    const int n_chunks = (n_img + ref_batch - 1) / ref_batch;
    const int grp = std::max(1, c.max_batch / ref_batch);      // reference batches per encoder launch group
    int fb_first[2] = {-1, -1}, fb_count[2] = {0, 0};
    bool feat_used[2] = {false, false};
    int next_enc = 0;
    for (int ck = 0; ck < n_chunks;) {
        // keep both feature buffers busy on the encoder stream: the encoder of the following groups runs while the
        // beam search of these reference batches occupies the caller's stream
        for (int fb = 0; fb < 2; ++fb) {
            if (fb_first[fb] >= 0 || next_enc >= n_chunks) continue;
            const int cnt = std::min(grp, n_chunks - next_enc);
            const int first = next_enc * ref_batch, n = std::min(cnt * ref_batch, n_img - first);
            if (feat_used[fb]) HIPCHK(h, hipStreamWaitEvent(h->enc_stream, h->ev_feat_free[fb], 0));
            const int rc = mnx_encode(h, images + (size_t)first * img_elems, n, h->feat_ring[fb], h->enc_stream);
            if (rc != MNX_OK) return rc;
            HIPCHK(h, hipEventRecord(h->ev_enc_done[fb], h->enc_stream));
            fb_first[fb] = next_enc; fb_count[fb] = cnt; next_enc += cnt;
        }
        int fb = -1;
        for (int i = 0; i < 2; ++i)
            if (fb_first[i] >= 0 && ck >= fb_first[i] && ck < fb_first[i] + fb_count[i]) fb = i;
        if (fb < 0) { h->err = "mnx_predict_beam: internal: reference batch without features"; return MNX_ERR_HIP; }
        // the next G reference batches of this feature buffer, searched together
        const int G = std::min(g_max, fb_first[fb] + fb_count[fb] - ck);
        const int first = ck * ref_batch, n = std::min(G * ref_batch, n_img - first);
        HIPCHK(h, hipStreamWaitEvent(s, h->ev_enc_done[fb], 0));
        const float* feats[MAX_BEAM_IMGS];
        for (int g = 0; g < G; ++g) feats[g] = h->feat_ring[fb] + (size_t)(ck + g - fb_first[fb]) * ref_batch * S * h->dw.enc_dim;
        int32_t* tok = tokens + (size_t)first * max_len;
        int rc = decode_beam_groups(h, feats, G, ref_batch, n, beam, 1, max_len, tok, lengths + first, scores + first, h->beam_hidden, s);
        if (rc != MNX_OK) return rc;
        ck += G;
        if (ck == fb_first[fb] + fb_count[fb]) {     // last reference batch of the group: the buffer is free again
            HIPCHK(h, hipEventRecord(h->ev_feat_free[fb], s));
            feat_used[fb] = true;
            fb_first[fb] = -1;
        }
        int32_t* aidx = atom_idx + (size_t)first * kmax;
        HIPCHK(h, atoms_enqueue_raw(h->tc_dev, tok, lengths + first, n, max_len, kmax, aidx, n_atoms + first, s));
        for (int o = 0; o < n; o += ROW_TILE) {      // the bond head's scratch holds one reference batch
            const int nb = std::min(ROW_TILE, n - o);
            HIPCHK(h, edges_enqueue(h->dw, h->db, h->beam_hidden + (size_t)o * max_len * D, nullptr, aidx + (size_t)o * kmax,
                                    n_atoms + first + o, nb, kmax, max_len, edges + (size_t)(first + o) * kmax * kmax, nullptr, s));
        }
    }
    HIPCHK(h, hipStreamSynchronize(s));
    return check_encoder_range(h, s);
}

int mnx_preprocess(mnx_engine* h, const uint8_t* rgb, int32_t height, int32_t width, int32_t pad,
                   int32_t pad_to_square, int32_t* crop_out, float* out, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!rgb || !out || height < 1 || width < 1 || pad < 0) { h->err = "mnx_preprocess: null/empty argument"; return MNX_ERR_INVALID_ARG; }
    if (height > 16384 || width > 16384 || pad > 4096) { h->err = "mnx_preprocess: image larger than 16384x16384"; return MNX_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, launch_preprocess(rgb, height, width, pad, pad_to_square ? 1 : 0, h->cfg.img_size, h->prep_bbox, crop_out, out,
                                (hipStream_t)stream));
    return MNX_OK;
}

int mnx_edges(mnx_engine* h, const float* hidden, const int32_t* atom_idx, const int32_t* n_atoms, int32_t B,
              int32_t kmax, int32_t max_len, uint8_t* edges, double* scores, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!hidden || !atom_idx || !n_atoms || !edges || B < 1 || kmax < 1 || max_len < 1) {
        h->err = "mnx_edges: null/empty argument";
        return MNX_ERR_INVALID_ARG;
    }
    if (B > ROW_TILE || kmax > h->db.kmax) { h->err = "mnx_edges: B <= 32 and kmax <= cfg.max_atoms required"; return MNX_ERR_CAPACITY; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, edges_enqueue(h->dw, h->db, hidden, nullptr, atom_idx, n_atoms, B, kmax, max_len, edges, scores, (hipStream_t)stream));
    return MNX_OK;
}

int mnx_set_token_classes(mnx_engine* h, const uint8_t* flags, int32_t n, int32_t lbracket, int32_t rbracket,
                          int32_t id_C, int32_t id_l, int32_t id_B, int32_t id_r) {
    if (!h || !flags || n < 1 || n > 256) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    TokenClasses tc{};
    memcpy(tc.flags, flags, (size_t)n);
    tc.lbracket = lbracket; tc.rbracket = rbracket; tc.id_C = id_C; tc.id_l = id_l; tc.id_B = id_B; tc.id_r = id_r;
    tc.x0 = h->cfg.sym_offset; tc.y0 = h->cfg.sym_offset + h->cfg.coord_bins; tc.vocab = h->cfg.vocab;
    HIPCHK(h, hipMemcpy(h->tc_dev, &tc, sizeof(tc), hipMemcpyHostToDevice));
    h->have_tc = true;
    return MNX_OK;
}

int mnx_atom_scan(mnx_engine* h, const int32_t* tokens, const int32_t* lengths, int32_t n, int32_t T, int32_t kmax,
                  int32_t* atom_idx, int32_t* n_atoms, void* stream) {
    if (!h || !tokens || !lengths || !atom_idx || !n_atoms || n < 1 || T < 1 || kmax < 1) return MNX_ERR_INVALID_ARG;
    if (!h->have_tc) { h->err = "mnx_atom_scan: call mnx_set_token_classes first"; return MNX_ERR_INVALID_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, atoms_enqueue_raw(h->tc_dev, tokens, lengths, n, T, kmax, atom_idx, n_atoms, (hipStream_t)stream));
    return MNX_OK;
}

// The whole hot path for n_img images with continuous batching (see include/molnextr_hip.h).
int mnx_predict(mnx_engine* h, const float* images, int32_t n_img, int32_t ref_batch, int32_t max_len,
                int32_t stop_on_eos, int32_t* tokens, int32_t* lengths, int32_t* n_atoms, int32_t* atom_idx,
                uint8_t* edges, int32_t kmax, void* stream) {
    if (!h) return MNX_ERR_INVALID_ARG;
    if (!images || !tokens || !lengths || !n_atoms || !atom_idx || !edges || n_img < 1) {
        h->err = "mnx_predict: null/empty argument";
        return MNX_ERR_INVALID_ARG;
    }
    if (!h->have_tc) { h->err = "mnx_predict: call mnx_set_token_classes first"; return MNX_ERR_INVALID_ARG; }
    const mnx_config& c = h->cfg;
    if (ref_batch < 1 || ref_batch > ROW_TILE || ref_batch > c.max_batch || max_len < 1 || max_len > c.max_len ||
        kmax < 1 || kmax > h->db.kmax) {
        h->err = "mnx_predict: ref_batch <= min(32, max_batch), max_len <= cfg.max_len, kmax <= cfg.max_atoms required";
        return MNX_ERR_CAPACITY;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    if (!s) {
        if (!h->own_stream) HIPCHK(h, hipStreamCreate(&h->own_stream));
        s = h->own_stream;
    }
    const int S = h->db.S, D = c.dec_dim, SL = h->db.slots;
    const size_t img_elems = (size_t)3 * c.img_size * c.img_size;
    const int n_chunks = (n_img + ref_batch - 1) / ref_batch;
    struct Chunk { int first, n, tag, admit_seq; std::vector<int> slots; };
    std::vector<Chunk> live;
    std::vector<int> free_tags;      // a chunk tag is also its 32-slot row tile and its memory K/V block
    for (int i = h->n_chunk_bufs - 1; i >= 0; --i) free_tags.push_back(i);
    int* pinned = h->host_flag;                       // [2][1 + MAX_CHUNKS] snapshots, then slot lists
    int* pin_slots = h->host_flag + 2 * (1 + MAX_CHUNKS);
    int rc = MNX_OK;
    int bound = 0;                                    // upper bound of alive rows (host-side, conservative)
    std::vector<std::pair<int, int>> admits;          // (iteration, rows) of every admission
    HIPCHK(h, dec_enqueue_reset(h->db, s));
    // the range flag is per call: an earlier mnx_encode on a bad input (whose caller did not poll mnx_encoder_status) must
    // not make THIS job report MNX_ERR_RANGE
    HIPCHK(h, hipMemsetAsync(h->enc_flag, 0, sizeof(int), s));
    // the encoder stream must not start before the caller's stream reaches this point (images ready)
    HIPCHK(h, hipEventRecord(h->ev_poll[0], s));
    HIPCHK(h, hipStreamWaitEvent(h->enc_stream, h->ev_poll[0], 0));
    int next = 0, done = 0, seq = 0;
    const char* trace_path = getenv("MNX_TRACE");
    FILE* tf = trace_path ? fopen(trace_path, "a") : nullptr;
    struct ExitGuard {      // every exit path: nothing of this call may still be in flight, the trace file is closed
        mnx_engine* h; hipStream_t s; FILE*& tf;
        ~ExitGuard() {
            (void)hipStreamSynchronize(h->enc_stream);
            (void)hipStreamSynchronize(s);
            if (tf) { fclose(tf); tf = nullptr; }
        }
    } exit_guard{h, s, tf};
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_begin = now_ms();
    double host_wait_ms = 0.0;
    int next_enc = 0;                       // next chunk to hand to the encoder stream
    // the encoder is batch-invariant, so it runs on GROUPS of reference batches (as many as max_batch holds):
    // bigger GEMM grids, half the launches; each reference batch of the group is admitted on its own
    const int grp = std::max(1, c.max_batch / ref_batch);
    int fb_first[2] = {-1, -1}, fb_count[2] = {0, 0};   // chunks [first, first+count) live in feature buffer i
    bool feat_used[2] = {false, false};
    const int ticks_per_poll = 4;            // measured: 2-4 equal, 8 = -4 % (retirement lags)
    while (done < n_chunks) {
        // ---- encoder prefetch: keep both feature buffers busy on the encoder stream
        for (int fb = 0; fb < 2; ++fb) {
            if (fb_first[fb] >= 0 || next_enc >= n_chunks) continue;
            const int cnt = std::min(grp, n_chunks - next_enc);
            const int first = next_enc * ref_batch, n = std::min(cnt * ref_batch, n_img - first);
            if (feat_used[fb]) HIPCHK(h, hipStreamWaitEvent(h->enc_stream, h->ev_feat_free[fb], 0));
            rc = mnx_encode(h, images + (size_t)first * img_elems, n, h->feat_ring[fb], h->enc_stream);
            if (rc != MNX_OK) return rc;
            HIPCHK(h, hipEventRecord(h->ev_enc_done[fb], h->enc_stream));
            fb_first[fb] = next_enc;
            fb_count[fb] = cnt;
            next_enc += cnt;
        }
        // ---- admission (in image order): only once the chunk's features are READY, so the decode stream never
        //      waits for the encoder; project the memory and admit on the decode stream
        while (next < n_chunks && !free_tags.empty()) {
            int fb = -1;
            for (int i = 0; i < 2; ++i)
                if (fb_first[i] >= 0 && next >= fb_first[i] && next < fb_first[i] + fb_count[i]) fb = i;
            if (fb < 0) break;
            const int first = next * ref_batch, n = std::min(ref_batch, n_img - first);
            // nothing to decode: wait for the features instead of polling
            const bool idle = live.empty();
            hipError_t q = idle ? hipEventSynchronize(h->ev_enc_done[fb]) : hipEventQuery(h->ev_enc_done[fb]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) { h->err = std::string("encoder event: ") + hipGetErrorString(q); return MNX_ERR_HIP; }
            Chunk ck;
            ck.first = first; ck.n = n; ck.tag = free_tags.back(); ck.admit_seq = seq;
            free_tags.pop_back();
            for (int i = 0; i < n; ++i) ck.slots.push_back(ck.tag * ROW_TILE + i);
            HIPCHK(h, hipStreamWaitEvent(s, h->ev_enc_done[fb], 0));   // already complete: ordering only
            char* memkv = h->db.mem_kv + (size_t)ck.tag * ROW_TILE * c.dec_layers * 2 * c.dec_heads * kvq_block_bytes(h->db.Sq);
            const float* feats = h->feat_ring[fb] + (size_t)(next - fb_first[fb]) * ref_batch * S * h->dw.enc_dim;
            HIPCHK(h, launch_sgemm_tn(feats, h->dw.w_enc, h->dw.b_enc, h->db.memory, n * S, D, h->dw.enc_dim, s));
            HIPCHK(h, launch_sgemm_tn(h->db.memory, h->dw.w_memkv, h->dw.b_memkv, h->db.mem_kv32, n * S, c.dec_layers * 2 * D, D, s, S));
            HIPCHK(h, kvq_pack_enqueue(h->db.mem_kv32, memkv, n * c.dec_layers * 2 * c.dec_heads, S, h->db.Sq, s));
            if (next + 1 == fb_first[fb] + fb_count[fb]) {    // last reference batch of the group: buffer is free again
                HIPCHK(h, hipEventRecord(h->ev_feat_free[fb], s));
                feat_used[fb] = true;
                fb_first[fb] = -1;
            }
            int* sl_dev = h->slot_lists + (size_t)ck.tag * ROW_TILE;
            int* sl_pin = pin_slots + (size_t)ck.tag * ROW_TILE;     // pinned, private to this tag until it retires
            for (int i = 0; i < n; ++i) sl_pin[i] = ck.slots[i];
            HIPCHK(h, hipMemcpyAsync(sl_dev, sl_pin, (size_t)n * 4, hipMemcpyHostToDevice, s));
            HIPCHK(h, dec_enqueue_admit(h->db, sl_dev, nullptr, n, ck.tag, ck.tag * ROW_TILE, max_len, stop_on_eos ? 1 : 0, s));
            bound += n;
            admits.emplace_back(seq, n);
            live.push_back(std::move(ck));
            ++next;
        }
        if (live.empty()) continue;       // (only possible before the first admission)
        // ---- a group of ticks, then a status snapshot
        // launch the tick graph sized for the alive-row bound (dense active list: idle row tiles are not launched)
        // (one graph per capacity, captured on first use: multiples of 64 up to 1024 rows, of 128 up to 2048, of 256 beyond)
        const int cap_step = bound <= 1024 ? 64 : bound <= 2048 ? 128 : 256;
        const int rows_cap = std::min(SL, (std::max(bound, 1) + cap_step - 1) / cap_step * cap_step);
        // the begin kernel scans slot tiles 0 .. highest live tag only (tags are handed out lowest-first): a 20-batch job
        // keeps it to 1024 of the 3072 slots — one pass of its 1024 threads instead of three; steps of 1024 keep the number of
        // tick graphs (one per scan range and capacity) small
        int hi_tag = 0;
        for (const Chunk& ck : live) hi_tag = std::max(hi_tag, ck.tag);
        const int scan = std::min(SL, std::max(((hi_tag + 1) * ROW_TILE + 1023) / 1024 * 1024, rows_cap));
        hipGraphExec_t exec = nullptr;
        rc = get_tick_graph(h, scan, rows_cap, nullptr, 0, s, &exec);
        if (rc != MNX_OK) return rc;
        rc = run_ticks(h, exec, scan, rows_cap, nullptr, 0, ticks_per_poll, s);
        if (rc != MNX_OK) return rc;
        HIPCHK(h, dec_enqueue_status(h->db, scan, s));
        int* snap = pinned + (seq & 1) * (1 + MAX_CHUNKS);
        HIPCHK(h, hipMemcpyAsync(snap, &h->db.st->n_active, (size_t)(1 + MAX_CHUNKS) * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipEventRecord(h->ev_poll[seq & 1], s));
        // ---- retire chunks that the PREVIOUS snapshot shows finished (the GPU keeps ticking meanwhile)
        if (seq > 0) {
            const int ps = seq - 1;
            const double tw0 = tf ? now_ms() : 0.0;
            HIPCHK(h, hipEventSynchronize(h->ev_poll[ps & 1]));
            if (tf) host_wait_ms += now_ms() - tw0;
            const int* sn = pinned + (ps & 1) * (1 + MAX_CHUNKS);
            {   // alive rows now <= alive rows in that snapshot + rows admitted after it was taken
                int after = 0;
                for (auto& a : admits) if (a.first > ps) after += a.second;
                bound = std::min(bound, sn[0] + after);
                while (!admits.empty() && admits.front().first <= ps) admits.erase(admits.begin());
            }
            for (size_t i = 0; i < live.size();) {
                Chunk& ck = live[i];
                if (ck.admit_seq <= ps && sn[1 + ck.tag] == 0) {
                    int* sl_dev = h->slot_lists + (size_t)ck.tag * ROW_TILE;
                    int* o_idx = atom_idx + (size_t)ck.first * kmax;
                    int* o_na = n_atoms + ck.first;
                    HIPCHK(h, gather_enqueue(h->db, sl_dev, ck.n, max_len, tokens + (size_t)ck.first * max_len,
                                             lengths + ck.first, nullptr, nullptr, s));
                    HIPCHK(h, atoms_enqueue(h->db, h->tc_dev, sl_dev, ck.n, kmax, o_idx, o_na, s));
                    HIPCHK(h, edges_enqueue(h->dw, h->db, h->db.hidden, sl_dev, o_idx, o_na, ck.n, kmax, h->db.T,
                                            edges + (size_t)ck.first * kmax * kmax, nullptr, s));
                    free_tags.push_back(ck.tag);
                    live.erase(live.begin() + i);
                    ++done;
                } else {
                    ++i;
                }
            }
        }
        if (tf) fprintf(tf, "%.3f seq %d live %zu next %d next_enc %d done %d free_tiles %zu\n", now_ms() - t_begin, seq,
                        live.size(), next, next_enc, done, free_tags.size());
        ++seq;
        if (seq > 200000) { h->err = "mnx_predict: watchdog (decode did not terminate)"; return MNX_ERR_HIP; }
    }
    HIPCHK(h, hipStreamSynchronize(s));
    if (tf) fprintf(tf, "%.3f end host_wait_ms %.3f\n", now_ms() - t_begin, host_wait_ms);
    return check_encoder_range(h, s);
}

int mnx_gemm_clock(mnx_engine* h, int32_t reset, double* mhz) {
    if (!h || !mhz) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, x3_clock_read(mhz, reset != 0));
    return MNX_OK;
}

int mnx_probe_mfma(mnx_engine* h, int32_t ms_target, double* tflops, double* mhz, void* stream) {
    if (!h || !tflops || !mhz || ms_target < 1 || ms_target > 2000) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    // 8 MFMAs of 16 cycles per iteration and wave, two waves per SIMD: ~256 cycles per iteration at ~1.9 GHz
    const int iters = (int)((double)ms_target * 1e-3 * 1.9e9 / 256.0);
    HIPCHK(h, mfma_probe(iters, (hipStream_t)stream, tflops, mhz));
    return MNX_OK;
}

int mnx_profile_enable(mnx_engine* h, int32_t enable) {
    if (!h) return MNX_ERR_INVALID_ARG;
    h->profiling = enable != 0;
    h->prof_stride = enable > 1 ? enable : 1;
    h->prof_calls = 0;
    h->prof_groups = 0;
    return MNX_OK;
}

int mnx_profile_read(mnx_engine* h, int32_t kind, double* ms_out, double* work_out, int64_t* launches_out) {
    if (!h) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    double ms = 0.0, work = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < h->ev_used; ++i) {
        if (h->ev_pool[i].kind != kind) continue;
        float t = 0.f;
        HIPCHK(h, hipEventSynchronize(h->ev_pool[i].b));
        HIPCHK(h, hipEventElapsedTime(&t, h->ev_pool[i].a, h->ev_pool[i].b));
        ms += t;
        work += h->ev_pool[i].work;
        ++n;
    }
    if (ms_out) *ms_out = ms;
    if (work_out) *work_out = work;
    if (launches_out) *launches_out = n;
    if (kind < 0) h->ev_used = 0;       // kind < 0: reset the pool (after the per-kind reads)
    return MNX_OK;
}

int mnx_probe_decode_attn(mnx_engine* h, int32_t rows, int32_t t, int32_t iters, double* self_ms, double* cross_ms,
                          void* stream) {
    if (!h || rows < 1 || rows > h->db.slots || t < 0 || t >= h->db.T || iters < 1) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto& e : ev) HIPCHK(h, hipEventCreate(&e));
    hipError_t err = dec_probe_attn(h->dw, h->db, rows, t, iters, ev, s);
    if (err == hipSuccess) err = hipStreamSynchronize(s);
    float a = 0.f, b = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&a, ev[0], ev[1]);
    if (err == hipSuccess) err = hipEventElapsedTime(&b, ev[2], ev[3]);
    for (auto& e : ev) hipEventDestroy(e);
    if (err != hipSuccess) { h->err = std::string("mnx_probe_decode_attn: ") + hipGetErrorString(err); return MNX_ERR_HIP; }
    if (self_ms) *self_ms = (double)a / iters;
    if (cross_ms) *cross_ms = (double)b / iters;
    return MNX_OK;
}

int mnx_gemm16(mnx_engine* h, int32_t epi, const void* A, const void* W, void* C, const float* bias, int32_t M,
               int32_t N, int32_t K, void* stream) {
    if (!h || !A || !W || !C) return MNX_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const int dt = dt_base(h->dt);
    if (!bias && N <= h->zero_bias_n) bias = h->zero_bias;
    if (epi & 0x100) {      // test aid: the persistent fp32-output kernel (gemm_res.hip) whatever the dispatch would choose
        epi &= 0xff;
        if (!gemm_res_supports(dt, epi, M, N, K)) { h->err = "mnx_gemm16: shape / epilogue not supported by gemm_res"; return MNX_ERR_INVALID_ARG; }
        HIPCHK(h, launch_gemm_res(dt, epi, A, W, (float*)C, bias, epi == EPI_RESID_F32 ? (const float*)C : nullptr, M, N, K,
                                  (hipStream_t)stream));
        return MNX_OK;
    }
    HIPCHK(h, launch_gemm16(dt, epi, A, W, C, bias, epi == EPI_RESID_F32 ? (const float*)C : nullptr, M, N, K,
                            (hipStream_t)stream));
    return MNX_OK;
}

int mnx_gemm16_split(mnx_engine* h, int32_t epi, const void* A, int64_t a_lo, const void* W, int64_t w_lo, float oscale,
                     void* C, int64_t c_lo, const float* bias, int32_t M, int32_t N, int32_t K, int32_t terms,
                     void* stream) {
    if (!h || !A || !W || !C) return MNX_ERR_INVALID_ARG;
    if (!dt_split(h->dt)) { h->err = "mnx_gemm16_split: the engine's compute_dtype is not a split mode"; return MNX_ERR_INVALID_ARG; }
    if (a_lo < 0 || w_lo < 0 || c_lo < 0) { h->err = "mnx_gemm16_split: negative plane offset"; return MNX_ERR_INVALID_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    if (!bias) {            // the kernels take a bias vector unconditionally: the engine's zero vector stands in
        if (N > h->zero_bias_n) { h->err = "mnx_gemm16_split: bias == NULL needs N <= 2 * the widest stage"; return MNX_ERR_CAPACITY; }
        bias = h->zero_bias;
    }
    if (terms < 1 || terms > 3 || (terms == 2 && h->dt != MNX_DT_F16X3)) {
        h->err = "mnx_gemm16_split: terms must be 1, 3 or (FP16X3 / FP16X3M only) 2";
        return MNX_ERR_INVALID_ARG;
    }
    SplitArgs sp;
    sp.a_lo = (size_t)a_lo; sp.w_lo = (size_t)w_lo; sp.c_lo = (size_t)c_lo; sp.oscale = oscale; sp.terms = terms;
    if (epi & 0x400) { sp.c_planes = 1; epi &= ~0x400; }      // 16-bit epilogues: the hi output plane only
    if (epi & 0x200) {      // test aid: the 128x128 kernel whatever the dispatch would choose
        epi &= 0xff;
        HIPCHK(h, launch_gemm16_tile128(h->dt, epi, A, W, C, bias, epi == EPI_RESID_F32 ? (const float*)C : nullptr, M, N, K,
                                        (hipStream_t)stream, &sp));
        return MNX_OK;
    }
    if (epi & 0x100) {      // test aid: gemm_res.hip whatever the dispatch would choose
        epi &= 0xff;
        if (!gemm_res_supports(h->dt, epi, M, N, K)) { h->err = "mnx_gemm16_split: not supported by gemm_res"; return MNX_ERR_INVALID_ARG; }
        HIPCHK(h, launch_gemm_res(h->dt, epi, A, W, (float*)C, bias, epi == EPI_RESID_F32 ? (const float*)C : nullptr,
                                  M, N, K, (hipStream_t)stream, &sp));
        return MNX_OK;
    }
    HIPCHK(h, launch_gemm16(h->dt, epi, A, W, C, bias, epi == EPI_RESID_F32 ? (const float*)C : nullptr, M, N,
                            K, (hipStream_t)stream, &sp));
    return MNX_OK;
}

}  // extern "C"
