// pframe_program.cpp -- the frame-level C boundary (include/dsvc.h: dsvc_pframe_program_load / _compress / _decompress / _destroy).
//
// Replaces, for a caller that has no Python: DeepSVC.compress (video_model.py:137-151) and DeepSVC.decompress (video_model.py:153-167) as
// test_video.py:342,349 call them -- one call per P-frame each.  A PROGRAM BLOB (written once by deepsvc_amd/export.py from a warmed model)
// holds the recorded launch sequences of the frame segments and of the decoder-scan levels with every pointer turned into a relocation,
// the packed weights / biases / entropy-model tables they point to, the sizes of the scan workspaces and of the workspace arenas, and the
// integer CDF tables of the four entropy models.  Loading allocates all of that on the device and resolves the relocations; coding a frame
// is then: patch the three caller-owned frame tensors into the op arrays, issue the segments with dsvc_pframe_segment_run, and run the
// host range coder between them exactly where the bitstream format puts it (image_model.py:206-207,221,253-254 / :260,273-274,288).
// Overlap, as in the Python layer: the motion strings are range-coded on a worker thread while the GPU runs prediction and the residual
// analysis; the residual decode chain (host range decoder <-> scan kernels, one synchronisation per level) runs on a worker thread and a side
// stream owned by the program, beside the motion chain and the prediction segment of the caller's stream.
#include <cstring>
#include <exception>
#include <future>
#include <new>
#include <memory>
#include <string>
#include <vector>
#include "common.h"

namespace {

enum { KIND_TENSOR = 0, KIND_ARENA0 = 1, KIND_ARENA1 = 2, KIND_SLOT = 3 };
enum { SLOT_REF = 0, SLOT_CUR, SLOT_SM_FEA, SLOT_FEATURE_IN, SLOT_FEATURE_OUT, SLOT_RECON, SLOT_WARPED, SLOT_PREDICT, SLOT_N };

#pragma pack(push, 1)
struct BlobHead {
    char magic[8];
    uint32_t version, precision, H, W, n_tensors, n_programs;
    uint32_t n_codecs, scan_op_size;
    uint64_t arena[2];
    uint32_t feature_channels, feature_esize;
    uint64_t payload_off;
};
struct BlobTensor { uint64_t nbytes; int64_t data_off; };
struct BlobProgram { char name[16]; uint32_t variant, n_ops, n_relocs, pad; uint64_t ops_off, relocs_off; };
struct BlobReloc { uint32_t field_off, kind, index, pad; uint64_t delta; };
struct BlobRef { uint32_t kind, index; uint64_t delta; };
struct BlobTable { uint32_t rows, stride; uint64_t cdf_off, len_off, off_off; };
struct BlobCodec { uint32_t M, Nz, zh, zw, d, yh, yw, ngroups; uint32_t groups[8]; BlobRef refs[4]; BlobTable tables[2]; };
#pragma pack(pop)
static_assert(sizeof(BlobHead) == 72 && sizeof(BlobProgram) == 48 && sizeof(BlobReloc) == 24 && sizeof(BlobCodec) == 192, "blob layout (deepsvc_amd/export.py)");

struct Program {
    std::vector<dsvc_scan_op> ops;
    struct Patch { uint32_t field_off, slot; uint64_t delta; };
    std::vector<Patch> caller;               // relocations of the caller-owned frame tensors: patched per call
    bool present = false;
};

struct HostTable {
    std::vector<int32_t> cdf, len, off;
    std::vector<uint16_t> lut;
    dsvc_cdf_table t{};
};

struct Codec {
    int M = 0, Nz = 0, zh = 0, zw = 0, d = 0, yh = 0, yw = 0;
    std::vector<int> groups;
    int32_t *staging = nullptr, *z_sym = nullptr, *idx_dev = nullptr, *sym_dev = nullptr;
    int64_t nz = 0, ny = 0, ns = 0;
    HostTable eb, gc;
    std::vector<Program> dec;                // dec0 .. dec<ngroups>
    int32_t *h_stage = nullptr, *h_idx = nullptr, *h_sym = nullptr;      // pinned
    std::vector<int32_t> z_index;            // channel index of every hyper-latent symbol (EntropyBottleneck.compress: index = channel)
};

}  // namespace

struct dsvc_pframe_program {
    int device = 0, H = 0, W = 0, precision = 0, feature_channels = 0, feature_esize = 0;
    std::vector<void*> tensors;
    std::vector<uint64_t> tensor_bytes;      // sizes of `tensors`, kept from load time: every relocation / codec reference is checked against them
    void* arena[2] = {nullptr, nullptr};
    uint64_t arena_bytes[2] = {0, 0};
    uint64_t slot_bytes[SLOT_N] = {0};       // extents of the frame tensors (caller-owned ones by contract of the call)
    void* owned[SLOT_N] = {nullptr};         // library-owned frame tensors: ref / cur (NHWC fp32), recon / warped / predict (NHWC fp32)
    Program seg[2][6];                       // [variant][E1, E2a, E2, D1, DR, D2]
    Codec codec[2];
    hipStream_t side = nullptr;              // the residual decode chain's stream
    hipEvent_t ev_stage = nullptr, ev_side = nullptr, ev_main = nullptr;
    ~dsvc_pframe_program() {
        (void)hipSetDevice(device);
        if (side) (void)hipStreamDestroy(side);
        for (hipEvent_t e : {ev_stage, ev_side, ev_main}) if (e) (void)hipEventDestroy(e);
        for (void* p : tensors) if (p) (void)hipFree(p);
        for (void* p : arena) if (p) (void)hipFree(p);
        for (void* p : owned) if (p) (void)hipFree(p);
        for (auto& c : codec) {
            for (int32_t* p : {c.h_stage, c.h_idx, c.h_sym}) if (p) (void)hipHostFree(p);
            if (!c.eb.lut.empty()) dsvc_cdf_forget_lut(c.eb.lut.data());
            if (!c.gc.lut.empty()) dsvc_cdf_forget_lut(c.gc.lut.data());
        }
    }
};

namespace {

const char* kSegNames[6] = {"E1", "E2a", "E2", "D1", "DR", "D2"};

#define HIP_OK(call, what)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { dsvc_set_error("%s: %s", what, hipGetErrorString(e_)); return DSVC_E_LAUNCH; } \
    } while (0)

// [off, off + len) lies inside [0, total): written so that no sum of blob-controlled 64-bit values can wrap
inline bool inside(uint64_t off, uint64_t len, uint64_t total) { return off <= total && len <= total - off; }

int load_table(const uint8_t* pay, uint64_t pay_bytes, const BlobTable& bt, HostTable* ht) {
    // rows / stride are bounded before they are multiplied (a CDF table of compressai has <= 64 .. 320 rows of <= a few thousand entries)
    DSVC_REQUIRE(bt.rows > 0 && bt.rows <= (1u << 20) && bt.stride > 1 && bt.stride <= (1u << 16), "pframe_program_load: CDF table of %u rows x %u entries", bt.rows, bt.stride);
    const uint64_t n = (uint64_t)bt.rows * bt.stride;
    DSVC_REQUIRE(inside(bt.cdf_off, n * 4, pay_bytes) && inside(bt.len_off, (uint64_t)bt.rows * 4, pay_bytes) &&
                 inside(bt.off_off, (uint64_t)bt.rows * 4, pay_bytes), "pframe_program_load: CDF table outside the blob");
    ht->cdf.resize(n); ht->len.resize(bt.rows); ht->off.resize(bt.rows); ht->lut.resize((size_t)bt.rows * 256);
    std::memcpy(ht->cdf.data(), pay + bt.cdf_off, n * 4);
    std::memcpy(ht->len.data(), pay + bt.len_off, (size_t)bt.rows * 4);
    std::memcpy(ht->off.data(), pay + bt.off_off, (size_t)bt.rows * 4);
    ht->t = dsvc_cdf_table{ht->cdf.data(), (int32_t)bt.rows, (int32_t)bt.stride, ht->len.data(), ht->off.data(), nullptr};
    if (int rc = dsvc_cdf_build_lut(&ht->t, ht->lut.data())) return rc;        // validates every row (tables come from a file)
    ht->t.lut = ht->lut.data();
    return DSVC_OK;
}

// copy the ops of one blob program, resolve its relocations against the loaded buffers, keep the caller-slot ones for per-call patching
int load_program(dsvc_pframe_program* P, const uint8_t* pay, uint64_t pay_bytes, const BlobProgram& bp, Program* out) {
    const uint64_t ob = (uint64_t)bp.n_ops * sizeof(dsvc_scan_op), rb = (uint64_t)bp.n_relocs * sizeof(BlobReloc);
    DSVC_REQUIRE(inside(bp.ops_off, ob, pay_bytes) && inside(bp.relocs_off, rb, pay_bytes), "pframe_program_load: program '%.16s' outside the blob", bp.name);
    out->ops.resize(bp.n_ops);
    std::memcpy(out->ops.data(), pay + bp.ops_off, ob);
    const BlobReloc* rel = (const BlobReloc*)(pay + bp.relocs_off);
    for (uint32_t i = 0; i < bp.n_relocs; ++i) {
        const BlobReloc& r = rel[i];
        DSVC_REQUIRE((uint64_t)r.field_off + 8 <= ob, "pframe_program_load: relocation outside the op array of '%.16s'", bp.name);
        char* field = (char*)out->ops.data() + r.field_off;
        uint64_t base = 0;
        if (r.kind == KIND_TENSOR) {
            DSVC_REQUIRE(r.index < P->tensors.size(), "pframe_program_load: tensor index %u out of range", r.index);
            DSVC_REQUIRE(r.delta < P->tensor_bytes[r.index], "pframe_program_load: relocation of '%.16s' points %llu bytes into tensor %u of %llu bytes", bp.name,
                         (unsigned long long)r.delta, r.index, (unsigned long long)P->tensor_bytes[r.index]);
            base = (uint64_t)(uintptr_t)P->tensors[r.index];
        } else if (r.kind == KIND_ARENA0 || r.kind == KIND_ARENA1) {
            base = (uint64_t)(uintptr_t)P->arena[r.kind - KIND_ARENA0];
            DSVC_REQUIRE(base != 0, "pframe_program_load: '%.16s' uses an arena of size 0", bp.name);
            DSVC_REQUIRE(r.delta < P->arena_bytes[r.kind - KIND_ARENA0], "pframe_program_load: relocation of '%.16s' outside its arena", bp.name);
        } else if (r.kind == KIND_SLOT) {
            DSVC_REQUIRE(r.index < SLOT_N, "pframe_program_load: frame-tensor slot %u unknown", r.index);
            DSVC_REQUIRE(r.delta < P->slot_bytes[r.index], "pframe_program_load: relocation of '%.16s' outside frame tensor %u", bp.name, r.index);
            if (P->owned[r.index]) base = (uint64_t)(uintptr_t)P->owned[r.index];
            else { out->caller.push_back(Program::Patch{r.field_off, r.index, r.delta}); continue; }
        } else {
            DSVC_REQUIRE(false, "pframe_program_load: relocation kind %u unknown", r.kind);
        }
        const uint64_t v = base + r.delta;
        std::memcpy(field, &v, 8);
    }
    out->present = true;
    return DSVC_OK;
}

int run_program(Program& pr, void* stream, void* const* caller) {
    for (const auto& pt : pr.caller) {
        if (!caller[pt.slot]) { dsvc_set_error("pframe: the program needs frame tensor %u, which this call did not provide", pt.slot); return DSVC_E_ARG; }
        const uint64_t v = (uint64_t)(uintptr_t)caller[pt.slot] + pt.delta;
        std::memcpy((char*)pr.ops.data() + pt.field_off, &v, 8);
    }
    return dsvc_pframe_segment_run(stream, pr.ops.data(), (int32_t)pr.ops.size());
}

// host half of one codec's encoder (image_model.py:206-207,221,237-254): symbols / indexes leave the device in one copy (stage_out, stream-ordered),
// then z and y are range-coded on the host (encode_host: after the copy's event)
int stage_out(Codec& c, hipStream_t st, hipEvent_t ev) {
    HIP_OK(hipMemcpyAsync(c.h_stage, c.staging, (size_t)(c.nz + 2 * c.ny) * 4, hipMemcpyDeviceToHost, st), "pframe_compress: copy of the symbol planes");
    HIP_OK(hipEventRecord(ev, st), "pframe_compress: hipEventRecord");
    return DSVC_OK;
}
int encode_host(Codec& c, hipEvent_t ev, uint8_t* out_y, int64_t cap_y, int64_t* n_y, uint8_t* out_z, int64_t cap_z, int64_t* n_z) {
    HIP_OK(hipEventSynchronize(ev), "pframe_compress: synchronize");
    void* enc = nullptr;
    if (int rc = dsvc_rans_encoder_create(&enc)) return rc;
    struct Guard { void* e; ~Guard() { dsvc_rans_encoder_destroy(e); } } guard{enc};
    if (int rc = dsvc_rans_encode_with_indexes(enc, c.h_stage, c.z_index.data(), c.nz, &c.eb.t)) return rc;
    int64_t n = dsvc_rans_flush(enc, out_z, cap_z);
    if (n < 0) return (int)n;
    *n_z = n;
    if (int rc = dsvc_rans_encode_with_indexes(enc, c.h_stage + c.nz, c.h_stage + c.nz + c.ny, c.ny, &c.gc.t)) return rc;
    n = dsvc_rans_flush(enc, out_y, cap_y);
    if (n < 0) return (int)n;
    *n_y = n;
    return DSVC_OK;
}
// a worker thread's result: the error text is thread-local, so it travels with the code
struct Outcome { int rc; std::string msg; };

// one codec's decode chain (image_model.py:259-297): z string -> z_hat -> level by level: index planes to the host, range decoder, symbols back
int decode_codec(Codec& c, hipStream_t st, const uint8_t* y, int64_t ny_bytes, const uint8_t* z, int64_t nz_bytes, void* const* caller) {
    {
        void* dz = nullptr;
        if (int rc = dsvc_rans_decoder_create(&dz, z, nz_bytes)) return rc;
        const int rc = dsvc_rans_decode_stream(dz, c.z_index.data(), c.nz, &c.eb.t, c.h_sym);
        dsvc_rans_decoder_destroy(dz);
        if (rc) return rc;
    }
    HIP_OK(hipMemcpyAsync(c.z_sym, c.h_sym, (size_t)c.nz * 4, hipMemcpyHostToDevice, st), "pframe_decompress: copy of the z symbols");
    HIP_OK(hipStreamSynchronize(st), "pframe_decompress: synchronize");          // h_sym is reused below
    if (int rc = run_program(c.dec[0], st, caller)) return rc;
    void* dy = nullptr;
    if (int rc = dsvc_rans_decoder_create(&dy, y, ny_bytes)) return rc;
    struct Guard { void* d; ~Guard() { dsvc_rans_decoder_destroy(d); } } guard{dy};
    for (size_t k = 0; k < c.groups.size(); ++k) {
        const int64_t m = (int64_t)c.groups[k] * c.ns;
        HIP_OK(hipMemcpyAsync(c.h_idx, c.idx_dev, (size_t)m * 4, hipMemcpyDeviceToHost, st), "pframe_decompress: copy of the index planes");
        HIP_OK(hipStreamSynchronize(st), "pframe_decompress: synchronize");      // the one sync point of this level
        if (int rc = dsvc_rans_decode_stream(dy, c.h_idx, m, &c.gc.t, c.h_sym)) return rc;
        HIP_OK(hipMemcpyAsync(c.sym_dev, c.h_sym, (size_t)m * 4, hipMemcpyHostToDevice, st), "pframe_decompress: copy of the symbols");
        HIP_OK(hipStreamSynchronize(st), "pframe_decompress: synchronize");      // h_sym is rewritten by the next level
        if (int rc = run_program(c.dec[k + 1], st, caller)) return rc;
    }
    return DSVC_OK;
}

}  // namespace

static int program_load_impl(const void* blob, int64_t nbytes, int32_t device, dsvc_pframe_program** out) {
    DSVC_REQUIRE(blob && out && nbytes >= (int64_t)sizeof(BlobHead), "pframe_program_load: null / short blob");
    const uint8_t* b = (const uint8_t*)blob;
    BlobHead h;
    std::memcpy(&h, b, sizeof(h));
    DSVC_REQUIRE(std::memcmp(h.magic, "DSVCPRG1", 8) == 0 && h.version == 1, "pframe_program_load: not a DSVC program blob (version 1)");
    DSVC_REQUIRE(h.scan_op_size == sizeof(dsvc_scan_op), "pframe_program_load: blob written for another ABI (op size %u, library %zu)", h.scan_op_size, sizeof(dsvc_scan_op));
    DSVC_REQUIRE(h.n_codecs == 2 && h.H % 64 == 0 && h.W % 64 == 0 && h.H > 0 && h.W > 0 && h.H <= 16384 && h.W <= 16384 && h.precision <= 1 &&
                 h.feature_channels >= 1 && h.feature_channels <= 4096 && (h.feature_esize == 2 || h.feature_esize == 4) &&
                 h.n_tensors <= (1u << 20) && h.n_programs <= (1u << 16), "pframe_program_load: bad header");
    const uint64_t tab = sizeof(BlobHead) + (uint64_t)h.n_tensors * sizeof(BlobTensor) + (uint64_t)h.n_programs * sizeof(BlobProgram) + 2 * sizeof(BlobCodec);
    DSVC_REQUIRE(h.payload_off >= tab && h.payload_off <= (uint64_t)nbytes, "pframe_program_load: truncated tables");
    const uint8_t* pay = b + h.payload_off;
    const uint64_t pay_bytes = (uint64_t)nbytes - h.payload_off;
    const BlobTensor* bt = (const BlobTensor*)(b + sizeof(BlobHead));
    const BlobProgram* bp = (const BlobProgram*)(bt + h.n_tensors);
    const BlobCodec* bc = (const BlobCodec*)(bp + h.n_programs);

    HIP_OK(hipSetDevice(device), "pframe_program_load: hipSetDevice");
    std::unique_ptr<dsvc_pframe_program> P(new dsvc_pframe_program());
    P->device = device; P->H = (int)h.H; P->W = (int)h.W; P->precision = (int)h.precision;
    P->feature_channels = (int)h.feature_channels; P->feature_esize = (int)h.feature_esize;
    P->tensors.assign(h.n_tensors, nullptr);
    P->tensor_bytes.assign(h.n_tensors, 0);
    for (uint32_t i = 0; i < h.n_tensors; ++i) {
        BlobTensor t; std::memcpy(&t, &bt[i], sizeof(t));
        DSVC_REQUIRE(t.nbytes > 0 && t.nbytes < (1ull << 36), "pframe_program_load: tensor %u of %llu bytes", i, (unsigned long long)t.nbytes);
        HIP_OK(hipMalloc(&P->tensors[i], t.nbytes), "pframe_program_load: hipMalloc");
        P->tensor_bytes[i] = t.nbytes;
        if (t.data_off < 0) { HIP_OK(hipMemset(P->tensors[i], 0, t.nbytes), "pframe_program_load: hipMemset"); }
        else {
            DSVC_REQUIRE(inside((uint64_t)t.data_off, t.nbytes, pay_bytes), "pframe_program_load: tensor %u outside the blob", i);
            HIP_OK(hipMemcpy(P->tensors[i], pay + t.data_off, t.nbytes, hipMemcpyHostToDevice), "pframe_program_load: upload");
        }
    }
    for (int k = 0; k < 2; ++k)
        if (h.arena[k]) {
            DSVC_REQUIRE(h.arena[k] < (1ull << 38), "pframe_program_load: arena %d of %llu bytes", k, (unsigned long long)h.arena[k]);
            HIP_OK(hipMalloc(&P->arena[k], h.arena[k]), "pframe_program_load: hipMalloc (arena)");
            P->arena_bytes[k] = h.arena[k];
        }
    const size_t frame3 = (size_t)h.H * h.W * 3 * 4;
    for (int s : {SLOT_REF, SLOT_CUR, SLOT_RECON, SLOT_WARPED, SLOT_PREDICT}) {
        HIP_OK(hipMalloc(&P->owned[s], frame3), "pframe_program_load: hipMalloc (frame)");
        P->slot_bytes[s] = frame3;
    }
    // caller-owned frame tensors (include/dsvc.h): sm_fea float32 [1,256,H/4,W/4], the carried feature [1,C,H,W] in the storage type
    P->slot_bytes[SLOT_SM_FEA] = (uint64_t)256 * (h.H / 4) * (h.W / 4) * 4;
    P->slot_bytes[SLOT_FEATURE_IN] = P->slot_bytes[SLOT_FEATURE_OUT] = (uint64_t)h.H * h.W * h.feature_channels * h.feature_esize;

    for (int ci = 0; ci < 2; ++ci) {
        BlobCodec c; std::memcpy(&c, &bc[ci], sizeof(c));
        Codec& C = P->codec[ci];
        DSVC_REQUIRE(c.ngroups >= 1 && c.ngroups <= 8 && c.M > 0 && c.M <= 4096 && c.Nz > 0 && c.Nz <= 4096 && c.d > 0 && c.M % c.d == 0 &&
                     c.zh > 0 && c.zw > 0 && c.yh > 0 && c.yw > 0 && c.zh <= 4096 && c.zw <= 4096 && c.yh <= 4096 && c.yw <= 4096, "pframe_program_load: codec %d geometry", ci);
        C.M = c.M; C.Nz = c.Nz; C.zh = c.zh; C.zw = c.zw; C.d = c.d; C.yh = c.yh; C.yw = c.yw;
        C.nz = (int64_t)c.Nz * c.zh * c.zw; C.ny = (int64_t)c.M * c.yh * c.yw; C.ns = (int64_t)c.d * c.yh * c.yw;
        int gmax = 0, gsum = 0;
        for (uint32_t g = 0; g < c.ngroups; ++g) DSVC_REQUIRE(c.groups[g] >= 1 && c.groups[g] <= 64, "pframe_program_load: codec %d slice group of %u slices", ci, c.groups[g]);
        for (uint32_t g = 0; g < c.ngroups; ++g) { C.groups.push_back((int)c.groups[g]); gmax = std::max(gmax, (int)c.groups[g]); gsum += (int)c.groups[g]; }
        DSVC_REQUIRE(gsum * c.d == c.M, "pframe_program_load: codec %d slice groups do not cover the latent", ci);
        int32_t** dst[4] = {&C.staging, &C.z_sym, &C.idx_dev, &C.sym_dev};
        // what the staging copies of compress / decompress read and write behind each reference (stage_out, decode_codec), in int32 elements
        const uint64_t need[4] = {(uint64_t)(C.nz + 2 * C.ny), (uint64_t)C.nz, (uint64_t)gmax * (uint64_t)C.ns, (uint64_t)gmax * (uint64_t)C.ns};
        for (int r = 0; r < 4; ++r) {
            DSVC_REQUIRE(c.refs[r].kind == KIND_TENSOR && c.refs[r].index < h.n_tensors, "pframe_program_load: codec %d buffer reference", ci);
            DSVC_REQUIRE(inside(c.refs[r].delta, need[r] * 4, P->tensor_bytes[c.refs[r].index]),
                         "pframe_program_load: codec %d buffer %d (%llu bytes at +%llu) does not fit tensor %u of %llu bytes", ci, r, (unsigned long long)(need[r] * 4),
                         (unsigned long long)c.refs[r].delta, c.refs[r].index, (unsigned long long)P->tensor_bytes[c.refs[r].index]);
            *dst[r] = (int32_t*)((char*)P->tensors[c.refs[r].index] + c.refs[r].delta);
        }
        if (int rc = load_table(pay, pay_bytes, c.tables[0], &C.eb)) return rc;
        if (int rc = load_table(pay, pay_bytes, c.tables[1], &C.gc)) return rc;
        DSVC_REQUIRE(C.eb.t.rows == C.Nz, "pframe_program_load: codec %d: %d factorised-prior rows for %d channels", ci, C.eb.t.rows, C.Nz);
        C.z_index.resize((size_t)C.nz);
        for (int ch = 0; ch < C.Nz; ++ch)
            for (int64_t i = 0; i < (int64_t)c.zh * c.zw; ++i) C.z_index[(size_t)ch * c.zh * c.zw + i] = ch;
        HIP_OK(hipHostMalloc((void**)&C.h_stage, (size_t)(C.nz + 2 * C.ny) * 4), "pframe_program_load: hipHostMalloc");
        HIP_OK(hipHostMalloc((void**)&C.h_idx, (size_t)std::max<int64_t>(gmax * C.ns, C.nz) * 4), "pframe_program_load: hipHostMalloc");
        HIP_OK(hipHostMalloc((void**)&C.h_sym, (size_t)std::max<int64_t>(gmax * C.ns, C.nz) * 4), "pframe_program_load: hipHostMalloc");
        C.dec.resize(c.ngroups + 1);
    }
    for (uint32_t i = 0; i < h.n_programs; ++i) {
        BlobProgram p; std::memcpy(&p, &bp[i], sizeof(p));
        p.name[15] = 0;
        Program* dst = nullptr;
        if (p.variant <= 1) {
            for (int s = 0; s < 6; ++s) if (std::strcmp(p.name, kSegNames[s]) == 0) dst = &P->seg[p.variant][s];
        } else if (p.name[0] == 'c' && (p.name[1] == '0' || p.name[1] == '1') && std::strncmp(p.name + 2, ".dec", 4) == 0) {
            const int ci = p.name[1] - '0', k = std::atoi(p.name + 6);
            if (k >= 0 && k < (int)P->codec[ci].dec.size()) dst = &P->codec[ci].dec[k];
        }
        DSVC_REQUIRE(dst != nullptr, "pframe_program_load: program '%s' (variant %u) not understood", p.name, p.variant);
        if (int rc = load_program(P.get(), pay, pay_bytes, p, dst)) return rc;
    }
    for (int v = 0; v < 2; ++v)
        for (int s = 0; s < 6; ++s) DSVC_REQUIRE(P->seg[v][s].present, "pframe_program_load: segment %s (variant %d) missing from the blob", kSegNames[s], v);
    for (int ci = 0; ci < 2; ++ci)
        for (auto& d : P->codec[ci].dec) DSVC_REQUIRE(d.present, "pframe_program_load: a decoder level of codec %d is missing from the blob", ci);
    HIP_OK(hipStreamCreateWithFlags(&P->side, hipStreamNonBlocking), "pframe_program_load: hipStreamCreate");
    for (hipEvent_t* e : {&P->ev_stage, &P->ev_side, &P->ev_main}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming), "pframe_program_load: hipEventCreate");
    HIP_OK(hipDeviceSynchronize(), "pframe_program_load: synchronize");
    *out = P.release();
    return DSVC_OK;
}

extern "C" void dsvc_pframe_program_destroy(dsvc_pframe_program* p) { delete p; }

extern "C" int dsvc_pframe_program_info(const dsvc_pframe_program* p, dsvc_pframe_info* info) {
    DSVC_REQUIRE(p && info, "pframe_program_info: null argument");
    info->H = p->H; info->W = p->W; info->precision = p->precision;
    info->feature_bytes = (int64_t)p->H * p->W * p->feature_channels * p->feature_esize;
    // a range-coded string never exceeds 4 bytes per pushed item + the 8-byte state; escapes add at most 11 items per symbol (rans_host.cpp)
    for (int ci = 0; ci < 2; ++ci) {
        info->string_cap[2 * ci] = 4 * 12 * p->codec[ci].ny + 16;
        info->string_cap[2 * ci + 1] = 4 * 12 * p->codec[ci].nz + 16;
    }
    return DSVC_OK;
}

static int compress_impl(dsvc_pframe_program* P, void* stream, const float* ref, const float* cur, const float* sm_fea, const void* feature,
                         uint8_t* const out[4], const int64_t cap[4], int64_t nbytes[4]) {
    DSVC_REQUIRE(P && ref && cur && sm_fea && out && cap && nbytes, "pframe_compress: null argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipSetDevice(P->device), "pframe_compress: hipSetDevice");
    const int v = feature ? 1 : 0;
    void* caller[SLOT_N] = {nullptr};
    caller[SLOT_SM_FEA] = (void*)sm_fea; caller[SLOT_FEATURE_IN] = (void*)feature;
    // the reference's tensors are NCHW fp32 (test_video.py:342): one LDS-tiled transpose each into the library's NHWC copies
    if (int rc = dsvc_nchw_to_nhwc(stream, ref, P->owned[SLOT_REF], 3, DSVC_F32, 1, 3, P->H, P->W)) return rc;
    if (int rc = dsvc_nchw_to_nhwc(stream, cur, P->owned[SLOT_CUR], 3, DSVC_F32, 1, 3, P->H, P->W)) return rc;
    if (int rc = run_program(P->seg[v][0], stream, caller)) return rc;                                              // E1
    if (int rc = stage_out(P->codec[0], st, P->ev_stage)) return rc;
    // the motion strings are coded on a worker thread while the GPU runs prediction + the residual analysis (E2a, E2)
    const int dev = P->device;
    auto mv_job = std::async(std::launch::async, [P, dev, out, cap, nbytes]() {
        (void)hipSetDevice(dev);
        const int rc = encode_host(P->codec[0], P->ev_stage, out[0], cap[0], &nbytes[0], out[1], cap[1], &nbytes[1]);
        return Outcome{rc, rc ? std::string(dsvc_last_error()) : std::string()};
    });
    int rc = run_program(P->seg[v][1], stream, caller);                                                             // E2a
    if (!rc) rc = run_program(P->seg[v][2], stream, caller);                                                        // E2
    if (!rc) rc = stage_out(P->codec[1], st, P->ev_main);
    if (!rc) rc = encode_host(P->codec[1], P->ev_main, out[2], cap[2], &nbytes[2], out[3], cap[3], &nbytes[3]);      // residual strings
    const Outcome mv = mv_job.get();
    if (mv.rc) { dsvc_set_error("%s", mv.msg.c_str()); return mv.rc; }
    return rc;
}

static int decompress_impl(dsvc_pframe_program* P, void* stream, const float* ref, const uint8_t* const strings[4], const int64_t nbytes[4],
                           const float* sm_fea, const void* feature, void* feature_out, float* recon, float* warped, float* predict) {
    DSVC_REQUIRE(P && ref && strings && nbytes && sm_fea && feature_out && recon, "pframe_decompress: null argument");
    for (int i = 0; i < 4; ++i) DSVC_REQUIRE(strings[i] != nullptr && nbytes[i] >= 8, "pframe_decompress: string %d missing", i);
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(hipSetDevice(P->device), "pframe_decompress: hipSetDevice");
    const int v = feature ? 1 : 0;
    void* caller[SLOT_N] = {nullptr};
    caller[SLOT_SM_FEA] = (void*)sm_fea; caller[SLOT_FEATURE_IN] = (void*)feature; caller[SLOT_FEATURE_OUT] = feature_out;
    if (int rc = dsvc_nchw_to_nhwc(stream, ref, P->owned[SLOT_REF], 3, DSVC_F32, 1, 3, P->H, P->W)) return rc;
    // The residual chain depends on its strings and its own scan workspace only: worker thread + the program's side stream, beside the motion
    // chain and the prediction segment of the caller's stream; the side stream first waits for what the caller had queued (the encoder of this
    // frame may still be reading the shared scan workspace when compress() and decompress() are called back to back).
    HIP_OK(hipEventRecord(P->ev_main, st), "pframe_decompress: hipEventRecord");
    HIP_OK(hipStreamWaitEvent(P->side, P->ev_main, 0), "pframe_decompress: hipStreamWaitEvent");
    const int dev = P->device, var = v;
    void* caller_copy[SLOT_N];
    std::memcpy(caller_copy, caller, sizeof(caller));
    auto res_job = std::async(std::launch::async, [P, dev, var, strings, nbytes, &caller_copy]() {
        (void)hipSetDevice(dev);
        int rc = decode_codec(P->codec[1], P->side, strings[2], nbytes[2], strings[3], nbytes[3], caller_copy);   // residual chain
        if (!rc) rc = run_program(P->seg[var][4], (void*)P->side, caller_copy);                                   // DR
        if (!rc && hipEventRecord(P->ev_side, P->side) != hipSuccess) { dsvc_set_error("pframe_decompress: hipEventRecord (side)"); rc = DSVC_E_LAUNCH; }
        return Outcome{rc, rc ? std::string(dsvc_last_error()) : std::string()};
    });
    int rc = decode_codec(P->codec[0], st, strings[0], nbytes[0], strings[1], nbytes[1], caller);                   // motion chain
    if (!rc) rc = run_program(P->seg[v][3], stream, caller);                                                        // D1
    const Outcome res = res_job.get();
    if (res.rc) { dsvc_set_error("%s", res.msg.c_str()); return res.rc; }
    if (rc) return rc;
    HIP_OK(hipStreamWaitEvent(st, P->ev_side, 0), "pframe_decompress: hipStreamWaitEvent");
    if ((rc = run_program(P->seg[v][5], stream, caller))) return rc;                                                // D2
    // decoded frames back in the reference's layout (video_model.py:167 returns NCHW tensors)
    struct { int slot; float* dst; } outs[3] = {{SLOT_RECON, recon}, {SLOT_WARPED, warped}, {SLOT_PREDICT, predict}};
    for (auto& o : outs)
        if (o.dst)
            if (int rc = dsvc_nhwc_to_nchw(stream, P->owned[o.slot], 3, DSVC_F32, o.dst, 1, 3, P->H, P->W)) return rc;
    return DSVC_OK;
}

// ---- the exported entry points: no C++ exception leaves the library ------------------------------------------------------------
// std::async can throw std::system_error, the vectors sized from blob fields std::bad_alloc / std::length_error: a C, Go or Rust host
// would see std::terminate.  Every entry point turns them into a status + dsvc_last_error() text.
#define DSVC_NOEXCEPT_CALL(where_, expr)                                                       \
    try { return (expr); }                                                                   \
    catch (const std::bad_alloc&) { dsvc_set_error("%s: out of host memory", where_); return DSVC_E_ARG; }       \
    catch (const std::exception& e) { dsvc_set_error("%s: %s", where_, e.what()); return DSVC_E_LAUNCH; }        \
    catch (...) { dsvc_set_error("%s: unknown C++ exception", where_); return DSVC_E_LAUNCH; }

extern "C" int dsvc_pframe_program_load(const void* blob, int64_t nbytes, int32_t device, dsvc_pframe_program** out) {
    DSVC_NOEXCEPT_CALL("pframe_program_load", program_load_impl(blob, nbytes, device, out))
}
extern "C" int dsvc_pframe_compress(dsvc_pframe_program* P, void* stream, const float* ref, const float* cur, const float* sm_fea, const void* feature,
                                    uint8_t* const out[4], const int64_t cap[4], int64_t nbytes[4]) {
    DSVC_NOEXCEPT_CALL("pframe_compress", compress_impl(P, stream, ref, cur, sm_fea, feature, out, cap, nbytes))
}
extern "C" int dsvc_pframe_decompress(dsvc_pframe_program* P, void* stream, const float* ref, const uint8_t* const strings[4], const int64_t nbytes[4],
                                      const float* sm_fea, const void* feature, void* feature_out, float* recon, float* warped, float* predict) {
    DSVC_NOEXCEPT_CALL("pframe_decompress", decompress_impl(P, stream, ref, strings, nbytes, sm_fea, feature, feature_out, recon, warped, predict))
}
