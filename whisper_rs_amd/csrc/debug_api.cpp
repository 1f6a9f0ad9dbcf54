// extern "C" test hooks (include/wrs_debug.h): not part of the drop-in boundary and NOT in the product library — this file is
// linked into libwrs_mi355x_debug.so only (the same objects as libwrs_mi355x.so plus this one), which tests and smoke() load.
#include "handles.h"
#include "../../include/wrs_debug.h"
#include <cmath>

using namespace wrs;

namespace {

// loading the test build turns the library's experiment switches on (common.h: exp_env)
const bool g_switches_on = (wrs::g_experiment_switches = true);

struct DebugCtx {
    hipStream_t s = nullptr;
};
DebugCtx& debug_ctx(int device_id) {
    static DebugCtx ctx[64];
    require_device(device_id);
    WRS_REQUIRE(device_id < 64, "device ordinal out of range");
    HIP_CHECK(hipSetDevice(device_id));
    if (!ctx[device_id].s) HIP_CHECK(hipStreamCreateWithFlags(&ctx[device_id].s, hipStreamNonBlocking));
    return ctx[device_id];
}

}  // namespace

extern "C" {

int wrs_debug_gemm_f16(wrs_debug_gemm* g) {
    return guarded([&] {
        WRS_REQUIRE(g && g->a && g->w && g->M > 0 && g->N > 0 && g->K > 0, "wrs_debug_gemm_f16: bad argument");
        DebugCtx& cx = debug_ctx(g->device_id);
        hipStream_t s = cx.s;
        // weights through the model's own packer (Model::finalize: rows padded to 128, K to 64)
        const int64_t n_pad = round_up(std::max<int64_t>(g->N, g->n_w), 128), k_pad = round_up(g->K, 64);
        WRS_REQUIRE(g->n_w >= 1 && g->n_w <= n_pad && g->k_w >= 1 && g->k_w <= k_pad, "wrs_debug_gemm_f16: weight shape [%ld, %ld] does not fit N=%ld K=%ld",
                    (long)g->n_w, (long)g->k_w, (long)g->N, (long)g->K);
        DevBuf wp;
        wp.alloc(packed_weight_halfs(n_pad, k_pad) * 2);
        if (g->conv_c > 0) {
            WRS_REQUIRE(g->k_w == 3 * (int64_t)g->conv_c, "wrs_debug_gemm_f16: conv weight needs k_w = 3 * conv_c");
            launch_pack_conv_weight_f32(g->w, g->n_w, g->conv_c, wp.as<half_t>(), n_pad, k_pad, s);
        } else {
            launch_pack_weight_f32(g->w, g->n_w, g->k_w, g->k_w, wp.as<half_t>(), n_pad, k_pad, s);
        }
        g->kernel_used = WRS_GEMM_AUTO;
        if (g->path == WRS_GEMM_PATH_TILED) {
            GemmA A;
            A.a = reinterpret_cast<const half_t*>(g->a);
            A.packed = g->a_packed;
            if (g->a_packed) {
                A.packed_rows = g->packed_rows;
            } else {
                A.lda = g->lda;
                if (g->rows_per_batch > 0) { A.rows_per_batch = g->rows_per_batch; A.batch_stride = g->batch_stride; }
            }
            GemmEpi e;
            e.bias = g->bias; e.scale = g->scale; e.scale_n_end = g->scale_n_end; e.act = g->act;
            e.residual = g->residual; e.posemb = g->posemb; e.c = g->c; e.c_f32 = g->c_f32; e.ldc = g->ldc;
            if (g->c_rows_per_batch > 0) { e.c_rows_per_batch = g->c_rows_per_batch; e.c_batch_rows = g->c_batch_rows; e.c_row_off = g->c_row_off; }
            e.c_packed = g->c_packed;
            e.ksplit = g->ksplit > 1 ? g->ksplit : 1;
            e.slab_stride = g->slab_stride;
            DevBuf sp;
            if (g->kcache) {  // the decode step's QKV-to-cache epilogue of the tiled kernel (what the benchmark runs at >= 480 rows)
                WRS_REQUIRE(g->vcache && g->n_ctx > 0 && g->cache_pos >= 0 && g->cache_pos < g->n_ctx && g->N % 3 == 0, "wrs_debug_gemm_f16: bad QKV cache arguments");
                sp.alloc(sizeof(StepParams));
                StepParams h{};
                h.cache_pos = g->cache_pos; h.n_rows = (int)g->M;
                HIP_CHECK(hipMemcpyAsync(sp.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));
                e.qkv_d_model = (int)(g->N / 3);
                e.kcache = reinterpret_cast<half_t*>(g->kcache); e.vcache = reinterpret_cast<half_t*>(g->vcache);
                e.cache_row_stride = (int64_t)g->n_ctx * (g->N / 3); e.sp = sp.as<StepParams>();
            }
            if (g->pack_k) {
                e.pack_k = reinterpret_cast<half_t*>(g->pack_k); e.pack_v = reinterpret_cast<half_t*>(g->pack_v);
                e.pack_T = g->pack_T; e.pack_H = g->pack_H; e.pack_k_col0 = g->pack_k_col0; e.pack_v_col0 = g->pack_v_col0;
            }
            gemm_debug_force(g->force_kernel);
            try {
                launch_gemm_f16(A, wp.as<half_t>(), g->M, g->N, g->K, e, s);
            } catch (...) {
                gemm_debug_force(GEMM_AUTO);
                throw;
            }
            gemm_debug_force(GEMM_AUTO);
            g->kernel_used = gemm_debug_last();
        } else if (g->path == WRS_GEMM_PATH_DEC) {
            WRS_REQUIRE(g->a_packed && g->c_f32 && g->c, "wrs_debug_gemm_f16: the decode GEMM takes packed A and writes fp32 slabs");
            const int m_pad = skinny_m_pad((int)g->M);
            WRS_REQUIRE(m_pad > 0 && g->packed_rows >= m_pad, "wrs_debug_gemm_f16: packed A too small for m_pad %d", m_pad);
            g->ksplit = gemm_dec_plan((int)g->M, g->N, g->K, nullptr, nullptr);
            g->slab_stride = (int64_t)m_pad * g->N;
            launch_gemm_dec(reinterpret_cast<const half_t*>(g->a), wp.as<half_t>(), (int)g->M, g->N, n_pad, g->K, reinterpret_cast<float*>(g->c), s);
        } else if (g->path == WRS_GEMM_PATH_SKINNY_F32) {
            WRS_REQUIRE(g->a_packed && g->c_f32 && g->c, "wrs_debug_gemm_f16: the skinny GEMM takes packed A and writes fp32");
            SkinnyEpi e;
            e.kind = SK_F32; e.c = g->c; e.ldc = g->ldc; e.bias = g->bias; e.act = g->act; e.scale = g->scale; e.scale_n_end = g->scale_n_end;
            launch_gemm_skinny(reinterpret_cast<const half_t*>(g->a), wp.as<half_t>(), (int)g->M, g->N, n_pad, g->K, 1, e, s);
        } else if (g->path == WRS_GEMM_PATH_PRING) {
            WRS_REQUIRE(g->a_packed && g->c, "wrs_debug_gemm_f16: the persistent ring GEMM takes packed A");
            PringEpi e;
            e.mode = g->pring_mode; e.bias = g->bias; e.scale = g->scale; e.scale_n_end = g->scale_n_end; e.ldc = g->ldc;
            if (g->pring_mode == WRS_PRING_RESID) e.x = reinterpret_cast<float*>(g->c);
            else e.c16 = reinterpret_cast<half_t*>(g->c);
            DevBuf sp;
            if (g->pring_mode == WRS_PRING_QKV) {
                WRS_REQUIRE(g->kcache && g->vcache && g->n_ctx > 0 && g->cache_pos >= 0 && g->cache_pos < g->n_ctx && g->phys_mul >= 1, "wrs_debug_gemm_f16: bad QKV cache arguments");
                sp.alloc(sizeof(StepParams));
                StepParams h{};
                h.cache_pos = g->cache_pos; h.n_rows = (int)g->M;
                HIP_CHECK(hipMemcpyAsync(sp.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));
                e.kcache = reinterpret_cast<half_t*>(g->kcache); e.vcache = reinterpret_cast<half_t*>(g->vcache);
                e.d_model = (int)(g->N / 3); e.cache_row_stride = (int64_t)g->n_ctx * (g->N / 3); e.phys_mul = g->phys_mul; e.sp = sp.as<StepParams>();
            }
            launch_gemm_pring(reinterpret_cast<const half_t*>(g->a), g->packed_rows, wp.as<half_t>(), g->M, g->N, g->K, g->pring_bm, e, s);
            HIP_CHECK(hipStreamSynchronize(s));
            g->pring_workgroups = pring_debug_last_wgs();
        } else {
            WRS_THROW(WRS_EUNSUPPORTED, "wrs_debug_gemm_f16: unknown path %d", g->path);
        }
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int wrs_debug_attn_self_run(wrs_debug_attn_self* a) {
    return guarded([&] {
        WRS_REQUIRE(a && a->qkv && a->qkv_bias && a->kc && a->vc && a->anc && a->out_packed, "wrs_debug_attn_self_run: null argument");
        WRS_REQUIRE(a->n_rows >= 1 && a->rows_per_audio >= 1 && a->n_rows % a->rows_per_audio == 0 && a->D == a->H * 64 && a->cache_pos >= 0 &&
                        a->cache_pos < a->n_ctx && a->n_ctx <= 512, "wrs_debug_attn_self_run: bad shape");
        DebugCtx& cx = debug_ctx(a->device_id);
        hipStream_t s = cx.s;
        const int R = a->n_rows;
        // step scalars + ancestry table in the layout the kernel reads: anc [2][R][n_ctx], parity 0 read by the kernel
        DevBuf sp, anc;
        sp.alloc(sizeof(StepParams));
        StepParams h{};
        h.cache_pos = a->cache_pos; h.pos_idx = a->cache_pos; h.parity = 0; h.n_rows = R;
        HIP_CHECK(hipMemcpyAsync(sp.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
        anc.alloc((size_t)2 * R * a->n_ctx * 4);
        HIP_CHECK(hipMemcpyAsync(anc.p, a->anc, (size_t)R * a->n_ctx * 4, hipMemcpyDeviceToDevice, s));
        if (a->q16_mode == 2 || a->q16_mode == 3) {  // beam-de-duplicating MFMA form (attn_self_mfma.hip): key list of this ancestry table first
            WRS_REQUIRE(a->rows_per_audio <= 8, "wrs_debug_attn_self_run: the MFMA form takes at most 8 rows per audio");
            const int n_audio = R / a->rows_per_audio, stride = a->rows_per_audio * a->n_ctx + 1;
            const bool direct = a->q16_mode == 3;  // q as fp16, new k / v already in the cache and in the key list
            DevBuf klist, kcount, qd;
            klist.alloc((size_t)n_audio * stride * 4);
            kcount.alloc((size_t)n_audio * 4);
            launch_self_keys(anc.as<int>(), (int64_t)R * a->n_ctx, a->n_ctx, n_audio, a->rows_per_audio, sp.as<StepParams>(), klist.as<int>(), stride,
                             kcount.as<int>(), direct, s);
            if (direct) {  // what the QKV GEMM's epilogue leaves: prepared on the host from the same qkv / bias / scale
                const int D = a->D;
                std::vector<float> qkv((size_t)R * 3 * D), bias((size_t)3 * D);
                HIP_CHECK(hipMemcpy(qkv.data(), a->qkv, qkv.size() * 4, hipMemcpyDeviceToHost));
                HIP_CHECK(hipMemcpy(bias.data(), a->qkv_bias, bias.size() * 4, hipMemcpyDeviceToHost));
                std::vector<half_t> q16((size_t)R * D), k16((size_t)D), v16((size_t)D);
                qd.alloc(q16.size() * 2);
                for (int r = 0; r < R; ++r) {
                    for (int c = 0; c < D; ++c) {
                        q16[(size_t)r * D + c] = (half_t)((qkv[(size_t)r * 3 * D + c] + bias[c]) * a->scale);
                        k16[c] = (half_t)((qkv[(size_t)r * 3 * D + D + c] + bias[D + c]) * a->scale);
                        v16[c] = (half_t)(qkv[(size_t)r * 3 * D + 2 * D + c] + bias[2 * D + c]);
                    }
                    const size_t off = ((size_t)r * a->n_ctx + a->cache_pos) * D * 2;
                    HIP_CHECK(hipMemcpy((char*)a->kc + off, k16.data(), (size_t)D * 2, hipMemcpyHostToDevice));
                    HIP_CHECK(hipMemcpy((char*)a->vc + off, v16.data(), (size_t)D * 2, hipMemcpyHostToDevice));
                }
                HIP_CHECK(hipMemcpy(qd.p, q16.data(), q16.size() * 2, hipMemcpyHostToDevice));
                launch_attn_self_mfma_q16(qd.as<half_t>(), reinterpret_cast<half_t*>(a->kc), reinterpret_cast<half_t*>(a->vc), a->n_ctx, R, a->rows_per_audio,
                                          a->H, a->D, sp.as<StepParams>(), klist.as<int>(), stride, kcount.as<int>(), reinterpret_cast<half_t*>(a->out_packed), s);
            } else {
                launch_attn_self_mfma(a->qkv, 1, R, a->qkv_bias, a->scale, reinterpret_cast<half_t*>(a->kc), reinterpret_cast<half_t*>(a->vc), a->n_ctx, R,
                                      a->rows_per_audio, a->H, a->D, sp.as<StepParams>(), klist.as<int>(), stride, kcount.as<int>(),
                                      reinterpret_cast<half_t*>(a->out_packed), s);
            }
            HIP_CHECK(hipStreamSynchronize(s));
            return;
        }
        if (a->q16_mode) {
            // the >= 480-row form: q as fp16 [R, D] and the new k / v already in the cache (what the QKV GEMM's epilogue leaves);
            // prepared on the host from the same qkv / bias / scale
            const int D = a->D;
            std::vector<float> qkv((size_t)R * 3 * D), bias((size_t)3 * D);
            HIP_CHECK(hipMemcpy(qkv.data(), a->qkv, qkv.size() * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(bias.data(), a->qkv_bias, bias.size() * 4, hipMemcpyDeviceToHost));
            std::vector<half_t> q16((size_t)R * D), k16((size_t)D), v16((size_t)D);
            DevBuf qd;
            qd.alloc(q16.size() * 2);
            for (int r = 0; r < R; ++r) {
                for (int c = 0; c < D; ++c) {
                    q16[(size_t)r * D + c] = (half_t)((qkv[(size_t)r * 3 * D + c] + bias[c]) * a->scale);
                    k16[c] = (half_t)((qkv[(size_t)r * 3 * D + D + c] + bias[D + c]) * a->scale);
                    v16[c] = (half_t)(qkv[(size_t)r * 3 * D + 2 * D + c] + bias[2 * D + c]);
                }
                const size_t off = ((size_t)r * a->n_ctx + a->cache_pos) * D * 2;
                HIP_CHECK(hipMemcpy((char*)a->kc + off, k16.data(), (size_t)D * 2, hipMemcpyHostToDevice));
                HIP_CHECK(hipMemcpy((char*)a->vc + off, v16.data(), (size_t)D * 2, hipMemcpyHostToDevice));
            }
            HIP_CHECK(hipMemcpy(qd.p, q16.data(), q16.size() * 2, hipMemcpyHostToDevice));
            launch_attn_self_q16(qd.as<half_t>(), reinterpret_cast<half_t*>(a->kc), reinterpret_cast<half_t*>(a->vc), anc.as<int>(),
                                 (int64_t)R * a->n_ctx, a->n_ctx, R, a->rows_per_audio, 1, a->H, a->D, sp.as<StepParams>(),
                                 reinterpret_cast<half_t*>(a->out_packed), s);
            HIP_CHECK(hipStreamSynchronize(s));
            return;
        }
        launch_attn_self(a->qkv, 1, R, a->qkv_bias, a->scale, reinterpret_cast<half_t*>(a->kc), reinterpret_cast<half_t*>(a->vc), anc.as<int>(),
                         (int64_t)R * a->n_ctx, a->n_ctx, R, a->rows_per_audio, 1, a->H, a->D, sp.as<StepParams>(),
                         reinterpret_cast<half_t*>(a->out_packed), s);
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

int wrs_debug_attn_cross_run(wrs_debug_attn_cross* a) {
    return guarded([&] {
        WRS_REQUIRE(a && a->q && a->q_bias && a->k && a->v && a->out_packed, "wrs_debug_attn_cross_run: null argument");
        WRS_REQUIRE(a->n_audio >= 1 && a->G >= 1 && a->G <= 16 && a->D == a->H * 64 && a->T >= 1, "wrs_debug_attn_cross_run: bad shape");
        DebugCtx& cx = debug_ctx(a->device_id);
        hipStream_t s = cx.s;
        const size_t per_head = packed_kv_halfs_per_head(a->T);
        DevBuf kp, vp;
        kp.alloc((size_t)a->n_audio * a->H * per_head * 2);
        vp.alloc((size_t)a->n_audio * a->H * per_head * 2);
        // K and V given as two row-major [n_audio*T, D] matrices: pack each with the library's packer (offsets 0)
        launch_pack_kv(reinterpret_cast<const half_t*>(a->k), a->D, 0, 0, a->n_audio, a->T, a->H, kp.as<half_t>(), vp.as<half_t>(), s);
        DevBuf kp2, vp2;
        kp2.alloc(kp.bytes);
        vp2.alloc(vp.bytes);
        launch_pack_kv(reinterpret_cast<const half_t*>(a->v), a->D, 0, 0, a->n_audio, a->T, a->H, kp2.as<half_t>(), vp2.as<half_t>(), s);
        const int R = a->n_audio * a->G;
        if (a->q16_mode) {  // the >= 480-row form: the query GEMM's fp16 output (bias and scale applied)
            const int D = a->D;
            std::vector<float> q((size_t)R * D), qb((size_t)D);
            HIP_CHECK(hipMemcpy(q.data(), a->q, q.size() * 4, hipMemcpyDeviceToHost));
            HIP_CHECK(hipMemcpy(qb.data(), a->q_bias, qb.size() * 4, hipMemcpyDeviceToHost));
            std::vector<half_t> q16(q.size());
            for (size_t i = 0; i < q.size(); ++i) q16[i] = (half_t)((q[i] + qb[i % D]) * a->q_scale);
            DevBuf qd;
            qd.alloc(q16.size() * 2);
            HIP_CHECK(hipMemcpy(qd.p, q16.data(), q16.size() * 2, hipMemcpyHostToDevice));
            launch_attn_cross_q16(qd.as<half_t>(), kp.as<half_t>(), vp2.as<half_t>(), a->n_audio, a->G, a->T, a->H, reinterpret_cast<half_t*>(a->out_packed), s);
            HIP_CHECK(hipStreamSynchronize(s));
            return;
        }
        launch_attn_cross(a->q, 1, R, a->q_bias, a->q_scale, kp.as<half_t>(), vp2.as<half_t>(), a->n_audio, a->G, a->T, a->H,
                          reinterpret_cast<half_t*>(a->out_packed), s);
        HIP_CHECK(hipStreamSynchronize(s));
    });
}

// teacher-forced decode of mel [n_audio,80,3000] (host): see DecodeTask::run_forced
int wrs_debug_decode_forced(wrs_decode_task* t, const float* mel, int n_audio, int n_steps, const int32_t* ftok, const int32_t* fsrc,
                            float* out_lp, int32_t* out_id, double* out_no_speech) {
    return guarded([&] {
        WRS_REQUIRE(t && mel, "null argument");
        Model& m = *t->t.m;
        m.use();
        m.load_mel(mel, n_audio, WRS_MEM_HOST, WRS_N_FRAMES, 0);
        t->t.run_forced(n_audio, n_steps, ftok, fsrc, out_lp, out_id, out_no_speech);
    });
}
int wrs_debug_decode_forced_features(wrs_decode_task* t, const float* feats, int n_audio, int n_steps, const int32_t* ftok,
                                     const int32_t* fsrc, float* out_lp, int32_t* out_id, double* out_no_speech) {
    return guarded([&] {
        WRS_REQUIRE(t && feats, "null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_debug_decode_forced_features: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        Model& m = *t->t.m;
        m.use();
        m.ensure_encoder_ws(n_audio);
        const size_t n = (size_t)n_audio * m.T * m.D;
        HIP_CHECK(hipMemcpyAsync(m.enc_ws.xa32.p, feats, n * 4, hipMemcpyHostToDevice, m.stream));
        launch_f32_to_f16(m.enc_ws.xa32.as<float>(), m.enc_ws.xa16.as<half_t>(), (int64_t)n, m.stream);
        HIP_CHECK(hipStreamSynchronize(m.stream));
        m.enc_ws.n_encoded = n_audio;
        ++m.enc_ws.epoch;
        t->t.run_forced(n_audio, n_steps, ftok, fsrc, out_lp, out_id, out_no_speech, true);
    });
}
int wrs_debug_decode_resync(wrs_decode_task* t, const float* mel, int n_audio, int n_forced, const int32_t* ftok, const int32_t* fsrc,
                            const float* fslp, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(t && mel && result && n_forced >= 0 && (n_forced == 0 || (ftok && fsrc && fslp)), "wrs_debug_decode_resync: bad argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_debug_decode_resync: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        DecodeTask& d = t->t;
        Model& m = *d.m;
        m.use();
        m.load_mel(mel, n_audio, WRS_MEM_HOST, WRS_N_FRAMES, 0);
        d.keep_candidates = true;
        d.forced_n = n_forced; d.forced_tok = ftok; d.forced_src = fsrc; d.forced_slp = fslp;
        try {
            d.run_loaded(n_audio, result);
        } catch (...) {
            d.forced_n = 0; d.forced_tok = d.forced_src = nullptr; d.forced_slp = nullptr;
            throw;
        }
        d.forced_n = 0; d.forced_tok = d.forced_src = nullptr; d.forced_slp = nullptr;
    });
}
int wrs_debug_decode_candidates(wrs_decode_task* t, int32_t* n_steps, int32_t* n_rows, int32_t* k, const float** lp, const int32_t** id) {
    return guarded([&] {
        WRS_REQUIRE(t && n_steps && n_rows && k && lp && id, "wrs_debug_decode_candidates: null argument");
        const DecodeTask& d = t->t;
        *k = d.mode == WRS_EXTRACT_BEAM_SEARCH ? d.group + 1 : 1;
        *n_rows = d.tr_rows;
        *n_steps = d.tr_rows > 0 ? (int32_t)(d.tr_cand_id.size() / ((size_t)d.tr_rows * *k)) : 0;
        *lp = d.tr_cand_lp.data();
        *id = d.tr_cand_id.data();
    });
}
// host sequencer alone (no device needed): see DecodeTask::simulate.  Trace of the run through out_* ([steps][rows]).
int wrs_debug_sequencer_run(const wrs_token_ids* ids, const wrs_decode_options* opt, const int64_t* initial, int32_t n_initial, int32_t n_audio,
                            int32_t n_steps, const float* lp, const int32_t* id, wrs_decode_result* out, int32_t* out_tokens,
                            int32_t* out_sources, float* out_slp) {
    return guarded([&] {
        WRS_REQUIRE(ids && opt && out, "null argument");
        DecodeTask t;
        t.simulate(*ids, *opt, initial, n_initial, n_audio, n_steps, lp, id, out);
        const size_t n = t.tr_tokens.size();
        if (out_tokens) memcpy(out_tokens, t.tr_tokens.data(), n * 4);
        if (out_sources) memcpy(out_sources, t.tr_src.data(), n * 4);
        if (out_slp) memcpy(out_slp, t.tr_slp.data(), n * 4);
    });
}
// long-form bookkeeping alone (no device needed): see simulate_transcribe
int wrs_debug_transcribe_bookkeeping(int32_t ts_begin, int64_t n_audio_ctxs, int64_t n_frames, const int64_t* prompt, int64_t n_prompt,
                                     const int64_t* window_tokens, const int32_t* window_lens, int32_t n_windows, const wrs_tokenizer* tok,
                                     wrs_transcribe_output* out) {
    return guarded([&] {
        simulate_transcribe(ts_begin, n_audio_ctxs, n_frames, prompt, n_prompt, window_tokens, window_lens, n_windows, tok ? &tok->t : nullptr, out);
    });
}
// f32 parity mode (csrc/parity_f32.hip): must be chosen before wrs_model_finalize
int wrs_debug_model_set_parity_f32(wrs_model* h, int enabled) {
    return guarded([&] {
        WRS_REQUIRE(h, "null model");
        WRS_REQUIRE(!h->m.finalized, "wrs_debug_model_set_parity_f32: the model is already finalized");
        h->m.parity_f32 = enabled != 0;
    });
}
int wrs_debug_model_step_paths(const wrs_model* h, int32_t* path) {
    return guarded([&] {
        WRS_REQUIRE(h && path, "null argument");
        *path = h->m.step_paths_seen;
    });
}
int wrs_debug_set_encoder_stop(wrs_model* h, int n_launches) { if (h) h->m.debug_enc_stop = n_launches; return 0; }
// copy an internal encoder workspace buffer to host as f32. which: 0 x(f32) 1 xn 2 qkv 3 att 4 h 5 c1 6 mel_t 7 xa16
int wrs_debug_read_encoder(wrs_model* h, int which, float* out, int64_t n) {
    return guarded([&] {
        WRS_REQUIRE(h && out && n > 0, "null argument");
        Model& m = h->m;
        m.use();
        HIP_CHECK(hipStreamSynchronize(m.stream));
        EncoderWorkspace& w = m.enc_ws;
        if (which == 0) { HIP_CHECK(hipMemcpy(out, w.x.p, (size_t)n * 4, hipMemcpyDeviceToHost)); return; }
        const DevBuf* b = which == 1 ? &w.xn : which == 2 ? &w.qkv : which == 3 ? &w.att : which == 4 ? &w.h : which == 5 ? &w.c1 : which == 6 ? &w.mel_t : &w.xa16;
        DevBuf tmp;
        tmp.alloc((size_t)n * 4);
        launch_f16_to_f32(b->as<half_t>(), tmp.as<float>(), n, m.stream);
        HIP_CHECK(hipStreamSynchronize(m.stream));
        HIP_CHECK(hipMemcpy(out, tmp.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
