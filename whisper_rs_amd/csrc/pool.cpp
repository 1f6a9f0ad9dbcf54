// Work queue over independent 30-s windows: the reference transcribes one window after the
// other on one device; windows cut at a fixed stride are independent units (decode/mod.rs:291-418 has no cross-audio
// term), so they shard over GPUs with no collective.  A pool owns one worker thread + one DecodeTask per model replica
// (replicas on different GPUs of the node, or several on one GPU); workers pull batches of up to `batch` consecutive
// windows from a shared counter and decode them in overlapped mode (the next batch's encoder is queued on the task's
// side stream before the current one is decoded).  Outputs are returned in window order.
#include "handles.h"
#include <atomic>
#include <mutex>
#include <thread>

namespace wrs {

struct WindowPool {
    std::vector<std::unique_ptr<DecodeTask>> tasks;
    int batch = 1;
    bool overlapped = true;

    // mel != null: windows of ONE recording's spectrogram [80, n_frames] (host), normalised over the whole file as the
    // reference does (transcribe.rs:111-112); window w = frames w*3000 .. (pad_or_trim, audio.rs:213-228)
    void run_audio(const float* audio, int64_t n_windows, int mem_kind, wrs_decode_result* result, const float* mel = nullptr,
                   int64_t n_frames = 0) {
        const bool is_mel = mel != nullptr;
        WRS_REQUIRE((audio || is_mel) && result && n_windows >= 1, "window pool: bad argument");
        if (mem_kind == WRS_MEM_DEVICE)
            for (auto& t : tasks) WRS_REQUIRE(t->m->device == tasks[0]->m->device, "window pool: a device buffer can only feed replicas on its own GPU");
        memset(result, 0, sizeof(*result));
        result->outputs = (wrs_decode_output*)calloc((size_t)n_windows, sizeof(wrs_decode_output));
        result->n_outputs = (int32_t)n_windows;
        const int64_t n_batches = cdiv(n_windows, batch);
        std::atomic<int64_t> next{0};
        std::mutex err_mutex;
        std::string err;
        int err_code = 0;
        std::atomic<int> max_steps{0};
        auto worker = [&](DecodeTask* t) {
            try {
                t->m->use();
                std::vector<float> mel_batch[2];  // host staging of spectrogram windows (two batches may be queued)
                int mel_slot = 0;
                auto span = [&](int64_t b, const float*& p, int& n) {
                    n = (int)std::min<int64_t>(batch, n_windows - b * batch);
                    if (!is_mel) { p = audio + b * batch * (int64_t)WRS_N_SAMPLES; return; }
                    std::vector<float>& buf = mel_batch[mel_slot];
                    mel_slot ^= 1;
                    buf.assign((size_t)n * 80 * WRS_N_FRAMES, 0.f);  // 0.0 in log-mel space past the end (audio.rs:226)
                    for (int i = 0; i < n; ++i) {
                        const int64_t f0 = (b * batch + i) * (int64_t)WRS_N_FRAMES;
                        const int64_t nf = std::min<int64_t>(WRS_N_FRAMES, n_frames - f0);
                        for (int c = 0; c < 80 && nf > 0; ++c)
                            memcpy(buf.data() + ((size_t)i * 80 + c) * WRS_N_FRAMES, mel + (size_t)c * n_frames + f0, (size_t)nf * 4);
                    }
                    p = buf.data();
                };
                const int kind = is_mel ? WRS_MEM_HOST : mem_kind;
                int64_t cur = next.fetch_add(1);
                if (cur >= n_batches) return;
                const float* p; int n;
                auto deliver = [&](int64_t b, wrs_decode_result& r) {
                    // hand the outputs over (the per-output buffers change owner; the array of this batch is dropped)
                    for (int i = 0; i < r.n_outputs; ++i) result->outputs[b * batch + i] = r.outputs[i];
                    free(r.outputs);
                    int ms = max_steps.load();
                    while (r.n_steps > ms && !max_steps.compare_exchange_weak(ms, r.n_steps)) {}
                };
                if (!overlapped) {  // one cross-K/V buffer per worker: encoder and decode of a batch back to back
                    for (; cur < n_batches; cur = next.fetch_add(1)) {
                        span(cur, p, n);
                        t->encode_async(p, n, kind, is_mel);
                        wrs_decode_result r;
                        t->run_pending(&r);
                        deliver(cur, r);
                    }
                    return;
                }
                span(cur, p, n);
                t->encode_async(p, n, kind, is_mel);
                while (cur < n_batches) {
                    const int64_t nxt = next.fetch_add(1);
                    if (nxt < n_batches) {
                        span(nxt, p, n);
                        t->encode_async(p, n, kind, is_mel);
                    }
                    wrs_decode_result r;
                    t->run_pending(&r);
                    deliver(cur, r);
                    cur = nxt;
                }
            } catch (const Error& e) {
                std::lock_guard<std::mutex> g(err_mutex);
                if (err.empty()) { err = e.what(); err_code = e.code; }
                next.store(n_batches);  // stop handing out work
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> g(err_mutex);
                if (err.empty()) { err = e.what(); err_code = WRS_EINVAL; }
                next.store(n_batches);
            }
        };
        std::vector<std::thread> threads;
        for (auto& t : tasks) threads.emplace_back(worker, t.get());
        for (auto& th : threads) th.join();
        // a failed worker may leave queued batches behind: drain them so the tasks stay usable
        for (auto& t : tasks) {
            while (!t->pending.empty()) {
                try {
                    t->m->use();
                    wrs_decode_result r;
                    t->run_pending(&r);
                    free_decode_result(&r);
                } catch (...) { t->pending.clear(); }
            }
        }
        result->n_steps = max_steps.load();
        if (!err.empty()) {
            free_decode_result(result);
            throw Error(err_code, err);
        }
    }
};

}  // namespace wrs

struct wrs_window_pool { wrs::WindowPool p; };

using namespace wrs;

extern "C" {

int wrs_window_pool_create(wrs_model* const* models, int32_t n_models, const wrs_token_ids* ids, const wrs_decode_options* opt,
                           int overlapped, wrs_window_pool** out) {
    return guarded([&] {
        WRS_REQUIRE(models && n_models >= 1 && n_models <= 64 && ids && opt && out, "wrs_window_pool_create: bad argument");
        std::unique_ptr<wrs_window_pool> h(new wrs_window_pool());
        h->p.batch = opt->max_audio > 0 ? opt->max_audio : 1;
        h->p.overlapped = overlapped != 0;
        for (int i = 0; i < n_models; ++i) {
            WRS_REQUIRE(models[i], "wrs_window_pool_create: null model");
            for (int j = 0; j < i; ++j) WRS_REQUIRE(models[j] != models[i], "wrs_window_pool_create: a model handle may back one worker only (a model owns one stream and one encoder workspace)");
            std::unique_ptr<DecodeTask> t(new DecodeTask());
            t->configure(&models[i]->m, *ids, *opt);
            h->p.tasks.push_back(std::move(t));
        }
        *out = h.release();
    });
}
int wrs_window_pool_run_audio(wrs_window_pool* p, const float* audio, int64_t n_windows, int mem_kind, wrs_decode_result* result) {
    return guarded([&] { WRS_REQUIRE(p, "null pool"); p->p.run_audio(audio, n_windows, mem_kind, result); });
}
int wrs_window_pool_run_mel(wrs_window_pool* p, const float* mel, int64_t n_frames, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(p && mel && n_frames >= 1, "wrs_window_pool_run_mel: bad argument");
        p->p.run_audio(nullptr, cdiv(n_frames, WRS_N_FRAMES), WRS_MEM_HOST, result, mel, n_frames);
    });
}
int wrs_window_pool_set_tokenizer(wrs_window_pool* p, const wrs_tokenizer* tok) {
    return guarded([&] {
        WRS_REQUIRE(p, "null pool");
        for (auto& t : p->p.tasks) t->tokenizer = tok ? &tok->t : nullptr;
    });
}
void wrs_window_pool_destroy(wrs_window_pool* p) {
    if (!p) return;
    for (auto& t : p->p.tasks) { (void)hipSetDevice(t->m->device); t.reset(); }
    delete p;
}

}  // extern "C"
