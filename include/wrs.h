/*
 * wrs.h — C-ABI of the MI355X-native Whisper hot path (libwrs_mi355x.so).
 *
 * Drop-in boundary for laptou/whisper-rs (reference, Rust).  In the reference every
 * tensor op crosses Rust->C through `tch`/`torch-sys` (Cargo.toml:16); this library
 * replaces that op layer one level coarser (mel / encoder / decoder step / filter+select /
 * window decode / transcribe) so that a `whisper` crate keeping the reference's public
 * signatures can bind it with `extern "C"` only — no tch types cross this boundary
 * (INTEGRATION.md shows the Rust side).  Each entry cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a WRS_E* code otherwise; wrs_last_error() gives
 *     the thread-local message (reference: anyhow::Result / panics; never unwinds here).
 *   - plain pointers and sizes only.  Host pointers unless a parameter says "device" (mem_kind = WRS_MEM_DEVICE).
 *   - DEVICE INPUTS MUST BE COMPLETE ON ENTRY: the library works on its own non-blocking HIP streams and does not order
 *     itself against the stream that produced a device buffer.  A caller that fills a device buffer with its own kernels
 *     or copies (e.g. torch) synchronises that work (hipStreamSynchronize / torch.cuda.synchronize()) before passing the
 *     pointer, and keeps the buffer unchanged until the call — for wrs_decode_task_encode_async until the matching
 *     wrs_decode_task_run_pending — has returned.  Device OUTPUTS are complete when a call returns.
 *   - handles are used from one thread at a time (reference types are !Send: Rc/RefCell).
 *   - a task holds a non-owning pointer to its model: the model must outlive its tasks
 *     (reference: DecodeTask<'a> borrows &'a Whisper, decode/mod.rs:24-25).
 *   - there is NO CPU fallback: every compute entry fails with WRS_ENODEVICE if no gfx950
 *     device is present.
 */
#ifndef WRS_H
#define WRS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WRS_OK 0
#define WRS_EINVAL 1     /* bad argument / shape (reference: debug_assert!/expect panics) */
#define WRS_ENODEVICE 2  /* no HIP device / wrong arch */
#define WRS_EHIP 3       /* HIP runtime error */
#define WRS_ESTATE 4     /* call sequence error (e.g. model not finalized, missing tensor) */
#define WRS_EUNSUPPORTED 5

/* audio constants — reference src/audio.rs:12-18 */
#define WRS_SAMPLE_RATE 16000
#define WRS_N_FFT 400
#define WRS_N_MELS 80
#define WRS_HOP_LENGTH 160
#define WRS_CHUNK_LENGTH 30
#define WRS_N_SAMPLES 480000
#define WRS_N_FRAMES 3000

const char* wrs_last_error(void);
const char* wrs_version(void);
/* number of usable gfx950 devices (0 if none); never fails. */
int wrs_device_count(void);

/* ---------------------------------------------------------------- model (src/model.rs) */

/* reference `ModelDims` (model.rs:546-557), same field order */
typedef struct wrs_model_dims {
    int64_t n_vocab;
    int64_t n_audio_ctxs;
    int64_t n_audio_states;
    int64_t n_audio_heads;
    int64_t n_audio_layers;
    int64_t n_text_ctxs;
    int64_t n_text_states;
    int64_t n_text_heads;
    int64_t n_text_layers;
} wrs_model_dims;

typedef struct wrs_model wrs_model;

#define WRS_F32 0
#define WRS_F16 1
#define WRS_MEM_HOST 0
#define WRS_MEM_DEVICE 1

/* `Whisper::new(vs.root(), dims)` (model.rs:560-584, examples/transcribe.rs:22-36). */
int wrs_model_create(const wrs_model_dims* dims, int device_id, wrs_model** out);
/* `vars.load("weights.ot")` (examples/transcribe.rs:37), one tensor at a time.  `name` is the
 * VarStore path (= OpenAI state_dict key); shape must match.  Values are
 * converted to fp16 for the MFMA path (LayerNorm/bias/positional tensors stay fp32). */
int wrs_model_set_tensor(wrs_model* m, const char* name, int dtype, const int64_t* shape, int ndim,
                         const void* data, int mem_kind);
/* `vars.load("weights.ot")` from a file: the same named tensors in safetensors format (F32, F16 or BF16; `.ot` is tch's
 * pickle-based archive, not read here — scripts/convert_checkpoint.py converts).  Tensors of other names are ignored;
 * *n_loaded (optional) = tensors taken.  Call wrs_model_finalize afterwards. */
int wrs_model_load_safetensors(wrs_model* m, const char* path, int32_t* n_loaded);
/* checks every tensor of the manifest is present and builds the fused/packed device layouts. */
int wrs_model_finalize(wrs_model* m);
/* `Whisper::device()` (model.rs:601-603): HIP device ordinal. */
int wrs_model_device(const wrs_model* m);
int wrs_model_get_dims(const wrs_model* m, wrs_model_dims* out);
void wrs_model_destroy(wrs_model* m);

/* `AudioEncoder::forward` (model.rs:417-441).  mel: [n_audio, 80, 3000] f32;
 * out: [n_audio, n_audio_ctxs, n_audio_states] f32. */
int wrs_encoder_forward(wrs_model* m, const float* mel, int n_audio, float* out);

/* `TextDecoder::forward_ext(tokens, xa, offset)` (model.rs:490-522) on the model's own K/V cache
 * (reference: the `Cached` cells inside the model, model.rs:37-98).
 *   tokens: [n_batch, n_tokens] i64; offset: positional-embedding offset of tokens[:,0];
 *   xa: [n_batch, n_audio_ctxs, n_text_states] f32, or NULL to reuse the cross-attention K/V
 *       computed by the previous call (the reference recomputes them on every call, model.rs:79;
 *       same values — SURVEY.md Q4);
 *   logits: [n_batch, n_tokens, n_vocab] f32.
 * New self-attention K/V are appended after the cached ones (Cached::forward, model.rs:83-87). */
int wrs_decoder_forward(wrs_model* m, const int64_t* tokens, int n_batch, int n_tokens, const float* xa,
                        int64_t offset, float* logits);
/* `Whisper::update_cache(&sources_indices)` (model.rs:591-594 -> :53-62): row r of every K/V cache
 * becomes row src[r]. */
int wrs_model_update_cache(wrs_model* m, const int64_t* src, int n_batch);
/* `Whisper::clear_cache()` (model.rs:596-599). */
int wrs_model_clear_cache(wrs_model* m);

/* ---------------------------------------------------------------- audio (src/audio.rs) */

/* `audio::load_audio(path)` (audio.rs:20-173) for RIFF/WAVE files (PCM 8/16/24/32-bit, IEEE float 32/64): decode,
 * downmix to mono by the channel mean (:121-127), resample to 16 kHz when the file's rate differs (:72-88).  The
 * reference resamples with rubato's FftFixedIn (third-party, not vendored): resampled output is band-limited to the
 * same Nyquist but not sample-identical (parity unpinned); 16 kHz files are bit-exact.  Other containers are out of
 * scope (WRS_EINVAL).  *out is library-owned: release with wrs_audio_free.  Host-only. */
int wrs_load_audio(const char* path, float** out, int64_t* n_samples);
/* the resampler alone: rate_in -> rate_out, n_out = ceil(n * rate_out / rate_in) */
int wrs_resample(const float* in, int64_t n, int32_t rate_in, int32_t rate_out, float** out, int64_t* n_out);
void wrs_audio_free(float* p);
/* `audio::mel_filter()` (audio.rs:175-188): out[80*201] row-major f32. */
int wrs_mel_filter(float* out);
/* `audio::pad_or_trim(x, len)` (audio.rs:213-228) on the last dim of a [rows, n] f32 array. */
int wrs_pad_or_trim(const float* x, int64_t rows, int64_t n, int64_t len, float* out);
/* `audio::log_mel_spectrogram(audio, mel_filter)` (audio.rs:230-259) on device `device_id`.
 *   audio: [n_batch, n_samples] f32 (each row an independent recording: its own global max,
 *   SURVEY.md Q11); filter: [80*201] f32 or NULL for wrs_mel_filter();
 *   out: [n_batch, 80, n_samples/160] f32.  mem_kind applies to audio and out. */
int wrs_log_mel_spectrogram(int device_id, const float* audio, int64_t n_batch, int64_t n_samples,
                            const float* filter, float* out, int mem_kind);

/* ---------------------------------------------------------------- tokenizer (src/tokenize.rs) */

/* special ids of reference `Tokenizer` (tokenize.rs:113-131) */
typedef struct wrs_token_ids {
    int32_t eot, sot, translate, transcribe, startoflm, startofprev, no_speech, no_timestamps, ts_begin;
    int32_t space; /* tokenizer.encode(" ")[0] (suppress_blanks.rs:22-28) */
} wrs_token_ids;

/* reference `Tokenizer` (tokenize.rs:111-131): GPT-2 byte-level BPE (`tokenizers::Tokenizer`) + Whisper special ids.
 * Host-only; usable without a GPU. */
typedef struct wrs_tokenizer wrs_tokenizer;

/* `Tokenizer::new_from_bytes(Task::Transcribe, bytes)` (tokenize.rs:142-202): bytes = a HuggingFace tokenizer.json
 * (model BPE, ByteLevel pre-tokenizer), e.g. the reference's tokenizer/gpt2.json.  The 106 Whisper specials are added
 * after the model vocabulary unless the file already holds them (a multilingual vocabulary then yields OpenAI's layout). */
int wrs_tokenizer_create_from_bytes(const void* bytes, size_t n_bytes, wrs_tokenizer** out);
/* `Tokenizer::new_from_file(task, path)` (tokenize.rs:204-207) */
int wrs_tokenizer_create_from_file(const char* path, wrs_tokenizer** out);
void wrs_tokenizer_destroy(wrs_tokenizer* t);
/* the `token_id_*` fields (tokenize.rs:113-131) + encode(" ")[0] */
int wrs_tokenizer_token_ids(const wrs_tokenizer* t, wrs_token_ids* out);
/* `tokenizer.encode(text, true)` via Deref (tokenize.rs:314-320; call sites transcribe.rs:131, suppress_blanks.rs:22).
 * utf8/n_bytes: the text (a Rust &str may hold NUL, so the length is explicit).
 * Writes up to cap ids; *n = number of ids the text has (call again with a larger buffer if *n > cap). */
int wrs_tokenizer_encode(const wrs_tokenizer* t, const char* utf8, size_t n_bytes, int32_t* out, int32_t cap, int32_t* n);
/* `Tokenizer::decode(&tokens)` (tokenize.rs:309-312): skip_special_tokens = false; ids outside the vocabulary
 * (timestamps) are dropped.  *out_text is library-owned, NUL-terminated UTF-8: release with wrs_string_free.
 * *out_len (optional) = its byte length, for the odd text that holds a NUL itself. */
int wrs_tokenizer_decode(const wrs_tokenizer* t, const int64_t* ids, int64_t n, char** out_text, size_t* out_len);
void wrs_string_free(char* s);
/* `Tokenizer::non_speech_tokens()` (tokenize.rs:215-307): sorted, duplicates kept, as the reference returns them */
int wrs_tokenizer_non_speech_tokens(const wrs_tokenizer* t, int32_t* out, int32_t cap, int32_t* n);
/* `tokenizer.token_to_id(token)` (tokenize.rs:170-197); -1 if absent */
int32_t wrs_tokenizer_token_to_id(const wrs_tokenizer* t, const char* token);
/* id of "<|xx|>" for a language code of the reference's LANGUAGES table (tokenize.rs:7-109); -1 if unknown */
int32_t wrs_tokenizer_language_token(const wrs_tokenizer* t, const char* code);
/* number of ids the tokenizer knows (model vocabulary + added specials) = ts_begin */
int32_t wrs_tokenizer_vocab_size(const wrs_tokenizer* t);

/* ---------------------------------------------------------------- decode (src/decode/) */

#define WRS_EXTRACT_GREEDY 0      /* TokenExtractMode::Greedy { group_sze } (extractor/mod.rs:12-16) */
#define WRS_EXTRACT_BEAM_SEARCH 1 /* TokenExtractMode::BeamSearch { beam_size, patience } */

/* reference `DecodeOptions` (decode/mod.rs:95-106); the prompt is set with wrs_decode_task_set_prompt */
typedef struct wrs_decode_options {
    int32_t sample_len;            /* <=0: None -> n_text_ctxs/2 (decode/mod.rs:129-131) */
    int32_t extract_mode;          /* WRS_EXTRACT_* */
    int32_t group_size;            /* beam_size, or Greedy group_sze */
    float patience;                /* beam search only */
    int32_t has_len_penalty;       /* Option<f64> */
    double len_penalty;
    int32_t has_max_initial_timestamp;
    double max_initial_timestamp;
    int32_t timestamps;
    int32_t suppress_blank;
    int32_t suppress_non_speech;
    const int32_t* suppress_tokens; /* Option<Vec<u32>>; may be NULL */
    int32_t n_suppress_tokens;
    const int32_t* non_speech_tokens; /* Tokenizer::non_speech_tokens() (tokenize.rs:215-307) */
    int32_t n_non_speech_tokens;
    int32_t position_offset_quirk;  /* 1 = reference behaviour: decoder offset = loop index (decode/mod.rs:244) */
    int32_t max_audio;              /* capacity: windows decoded per run() call (>=1) */
    /* `Tokenizer::sequence_sot()` (tokenize.rs:209-213) minus its leading sot: NULL/0 = the reference ([sot]);
     * OpenAI's multilingual form [sot, <|lang|>, <|transcribe|> or <|translate|>] is given as {lang, task}. */
    const int32_t* sot_sequence_tail;
    int32_t n_sot_sequence_tail;
} wrs_decode_options;

/* reference `DecodeOutput` (decode/mod.rs:109-119); text is produced by the caller's tokenizer */
typedef struct wrs_decode_output {
    int64_t* tokens; /* library-owned, n_tokens ids */
    int32_t n_tokens;
    double avg_logprob;
    double no_speech_prob;
    double sum_logprob;
    const float* audio_features_device; /* `DecodeOutput.audio_features` (decode/mod.rs:110): DEVICE pointer
                                         * [n_audio_ctxs, n_audio_states] fp32 into the model's encoder workspace, valid until
                                         * the next encoder pass on this model; NULL when a later batch was already queued with
                                         * wrs_decode_task_encode_async (overlapped mode / window pool).  Host copy:
                                         * wrs_decode_task_read_audio_features */
    char* text; /* `DecodeOutput.text` = tokenizer.decode(tokens) (decode/mod.rs:404): library-owned UTF-8, NULL unless a
                 * tokenizer is attached with wrs_decode_task_set_tokenizer */
} wrs_decode_output;

typedef struct wrs_decode_result {
    wrs_decode_output* outputs;
    int32_t n_outputs;
    int32_t n_steps; /* sampling steps taken */
} wrs_decode_result;

typedef struct wrs_decode_task wrs_decode_task;

/* `DecodeTask::new(&model, tokenizer, options)` (decode/mod.rs:122-204) */
int wrs_decode_task_create(wrs_model* m, const wrs_token_ids* ids, const wrs_decode_options* opt,
                           wrs_decode_task** out);
/* `DecodeTask::new` takes the tokenizer as Rc<Tokenizer> (decode/mod.rs:122-126): attach it to have `text` filled in.
 * Non-owning: the tokenizer must outlive the task.  NULL detaches. */
int wrs_decode_task_set_tokenizer(wrs_decode_task* t, const wrs_tokenizer* tok);
/* `DecodeTask::set_prompt(Option<&Tensor>)` (decode/mod.rs:206-218); n==0 or NULL => None */
int wrs_decode_task_set_prompt(wrs_decode_task* t, const int64_t* prompt, int32_t n);
/* `DecodeTask::run(&mel)` (decode/mod.rs:284-419).  mel: [n_audio, 80, 3000] f32 (the reference's
 * public run() takes one window; run_inner is batch-general, :291-347).  mem_kind: where mel lives. */
int wrs_decode_task_run(wrs_decode_task* t, const float* mel, int32_t n_audio, int mem_kind,
                        wrs_decode_result* result);
/* same, starting from encoder output [n_audio, n_audio_ctxs, n_state] f32 (run_inner after :296) */
int wrs_decode_task_run_features(wrs_decode_task* t, const float* audio_features, int32_t n_audio, int mem_kind,
                                 wrs_decode_result* result);
/* host copy of `DecodeOutput.audio_features` (decode/mod.rs:110,399) of window `audio_index` of the batch the task decoded
 * last: out[n_audio_ctxs * n_audio_states] f32, n_floats must be exactly that.  WRS_EINVAL once the encoder workspace has
 * been reused (another run, or wrs_decode_task_encode_async of a later batch). */
int wrs_decode_task_read_audio_features(wrs_decode_task* t, int32_t audio_index, float* out, int64_t n_floats);
/* same, starting from raw audio windows [n_audio, 480000] f32: log-mel + run.  EVERY WINDOW IS ITS OWN RECORDING here: the
 * log-mel floor (max - 8, audio.rs:256) uses the window's own maximum.  The reference's long-form loop normalises over the
 * whole file instead (transcribe.rs:111-112): wrs_transcribe_task_run / _run_chunked do that; this entry, encode_async and
 * the window pool are for windows that ARE separate recordings, or for callers that accept per-window normalisation
 * (the two give the same spectrogram unless a window's own maximum differs from the file's). */
int wrs_decode_task_run_audio(wrs_decode_task* t, const float* audio, int32_t n_audio, int mem_kind,
                              wrs_decode_result* result);
/* Overlapped mode (beyond the reference, whose loop is sequential: the unit that keeps a GPU's work queue full).
 * encode_async queues log-mel + AudioEncoder + cross-attention K/V projection of a batch of raw audio windows
 * ([n_audio, 480000] f32) on a side stream and returns at once; run_pending decodes the OLDEST queued batch.  Calling
 * encode_async for batch i+1 right before run_pending for batch i overlaps the MFMA-bound encoder of one batch with the
 * HBM-bound decode steps of the other (two cross-K/V buffers; at most two batches queued).  A device `audio` buffer must
 * stay valid until the batch has been decoded.  Results are identical to wrs_decode_task_run_audio. */
int wrs_decode_task_encode_async(wrs_decode_task* t, const float* audio, int32_t n_audio, int mem_kind);
int wrs_decode_task_run_pending(wrs_decode_task* t, wrs_decode_result* result);
/* same, on the windows whose encoder output is still resident from the previous run / detect call of this task's model */
int wrs_decode_task_run_encoded(wrs_decode_task* t, int32_t n_audio, wrs_decode_result* result);
/* Language identification (the reference declares `Task::LanguageId`, tokenize.rs:135-139, without implementing it;
 * semantics follow OpenAI whisper's detect_language): encoder, one decoder position with <|startoftranscript|>, softmax
 * over the n_languages ids starting at first_language_id (wrs_tokenizer_language_token(t, "en") ...).
 *   mel: [n_audio, 80, 3000] f32; out_language_ids: [n_audio]; out_probs: [n_audio, n_languages] or NULL. */
int wrs_decode_task_detect_language(wrs_decode_task* t, const float* mel, int32_t n_audio, int mem_kind,
                                    int32_t first_language_id, int32_t n_languages, int32_t* out_language_ids, float* out_probs);
/* per-window language token for the next runs: replaces sot_sequence_tail[0] for window a with language_ids[a]
 * (OpenAI decoding.py `tokens[:, sot_index + 1] = lang_tokens`).  n = 0 clears.  Needs n_sot_sequence_tail >= 1. */
int wrs_decode_task_set_languages(wrs_decode_task* t, const int32_t* language_ids, int32_t n);
/* per-step trace of the last run (test aid): for step s, row r: chosen token, source row, sum_logprob. */
int wrs_decode_task_trace(wrs_decode_task* t, int32_t* n_steps, int32_t* n_rows, const int32_t** tokens,
                          const int32_t** sources, const float** sum_logprobs);
void wrs_decode_result_free(wrs_decode_result* r);
void wrs_decode_task_destroy(wrs_decode_task* t);

/* the reference's internal plug points (decode/filter/mod.rs:12-14, decode/extractor/mod.rs:20-73),
 * exposed so each can be checked against the oracle in isolation:
 * fused SuppressBlank -> SuppressTokens -> TimestampTokens (decode/mod.rs:156-190) followed by the
 * extractor's log_softmax + top-k (beam_search.rs:37,62-65) on device.
 *   logits: [n_rows, n_vocab] f32 host; tokens: [n_rows, n_tokens] i64 host (history incl. prompt);
 *   sample_begin: prompt length; k <= 8;
 *   out_logprobs/out_ids: [n_rows, k]; out_filtered (optional): [n_rows, n_vocab] filtered logits. */
int wrs_filter_topk(wrs_decode_task* t, const float* logits, const int64_t* tokens, int32_t n_rows,
                    int32_t n_tokens, int32_t sample_begin, int32_t k, float* out_logprobs, int32_t* out_ids,
                    float* out_filtered);

/* ---------------------------------------------------------------- window pool (multi-GPU / multi-replica work queue) */

/* Fixed-stride 30-s windows are independent units (decode/mod.rs:291-418 has no cross-audio term; the reference itself
 * transcribes them one after the other, transcribe.rs:151-234): they shard over GPUs with no collective.  A pool runs
 * one worker thread + one decode task per model replica — replicas on different GPUs of the node (create one model per
 * device with wrs_model_create(dims, device_id, ..)), or several replicas on one GPU — and hands out batches of
 * opt->max_audio consecutive windows from a shared queue; every worker keeps its GPU busy in overlapped mode.
 *   audio: [n_windows, 480000] f32, host memory (or device memory when every replica sits on that GPU); each window is
 *   normalised as its own recording (see wrs_decode_task_run_audio; for whole-file normalisation compute the spectrogram
 *   once with wrs_log_mel_spectrogram and feed windows of it to wrs_decode_task_run); result->outputs[w] belongs to window w (same content as
 *   wrs_decode_task_run_audio on that window); release with wrs_decode_result_free.  The models must outlive the pool and
 *   must not be used elsewhere while a run is in progress. */
typedef struct wrs_window_pool wrs_window_pool;
/* overlapped != 0: every worker queues its next batch's encoder while it decodes (a second cross-K/V buffer per worker:
 * 2 x 2 x 1500 x n_text_states x n_text_layers x max_audio bytes more device memory); 0: encoder and decode back to back. */
int wrs_window_pool_create(wrs_model* const* models, int32_t n_models, const wrs_token_ids* ids, const wrs_decode_options* opt,
                           int overlapped, wrs_window_pool** out);
/* attach a tokenizer to every worker (outputs then carry `text`); non-owning, NULL detaches */
int wrs_window_pool_set_tokenizer(wrs_window_pool* p, const wrs_tokenizer* tok);
int wrs_window_pool_run_audio(wrs_window_pool* p, const float* audio, int64_t n_windows, int mem_kind, wrs_decode_result* result);
/* ONE long recording with the reference's whole-file normalisation: mel = wrs_log_mel_spectrogram of the whole file
 * ([80, n_frames] f32 host: global max, transcribe.rs:111-112), cut at the fixed 3000-frame stride (the reference's
 * `seek += N_FRAMES` branch, transcribe.rs:231; the last window zero-padded, audio.rs:213-228); result->outputs[w] is window
 * w.  Same tokens as wrs_transcribe_task_run_chunked on that recording, sharded over the pool's replicas. */
int wrs_window_pool_run_mel(wrs_window_pool* p, const float* mel, int64_t n_frames, wrs_decode_result* result);
void wrs_window_pool_destroy(wrs_window_pool* p);

/* ---------------------------------------------------------------- transcribe (src/transcribe.rs) */

/* reference `TranscribePrompt` (transcribe.rs:25-29); Text(String) is tokenized by the caller */
typedef struct wrs_transcribe_options {
    wrs_decode_options decode;
    const int64_t* prompt_tokens; /* Pretokenized(Vec<u32>) or NULL for None{..} */
    int32_t n_prompt_tokens;
    int32_t condition_on_prev_text; /* used when prompt_tokens == NULL */
} wrs_transcribe_options;

/* reference `TranscribeOutputSegment` (transcribe.rs:59-67) */
typedef struct wrs_segment {
    int64_t seek;
    double start_time, end_time;
    int64_t start_token, end_token;
    char* text; /* library-owned UTF-8 (transcribe.rs:200,227); NULL unless a tokenizer is attached */
} wrs_segment;

typedef struct wrs_transcribe_output {
    int64_t* tokens;
    int64_t n_tokens;
    wrs_segment* segments;
    int64_t n_segments;
    /* per decoded window (test aid) */
    int64_t* window_seek;
    double* window_avg_logprob;
    double* window_no_speech_prob;
    int64_t n_windows;
    char* text; /* `TranscribeOutput.text` = tokenizer.decode(tokens) (transcribe.rs:236); NULL unless a tokenizer is attached */
    /* 1: the loop stopped before the end of the audio — a window ended in a <|0.00|> timestamp pair (seek advance 0, on
     * which the reference loops forever, transcribe.rs:206) or 10 000 windows were decoded: the rest is NOT transcribed */
    int32_t truncated;
} wrs_transcribe_output;

typedef struct wrs_transcribe_task wrs_transcribe_task;

/* `TranscribeTask::new(&model, options)` (transcribe.rs:70-101) */
int wrs_transcribe_task_create(wrs_model* m, const wrs_token_ids* ids, const wrs_transcribe_options* opt,
                               wrs_transcribe_task** out);
/* `TranscribeOptions.tokenizer` (transcribe.rs:49): attach it to have text / segment text filled in.  Non-owning. */
int wrs_transcribe_task_set_tokenizer(wrs_transcribe_task* t, const wrs_tokenizer* tok);
/* `TranscribePrompt::Text(String)` (transcribe.rs:27,130-134): encoded with the attached tokenizer; implies conditioning. */
int wrs_transcribe_task_set_prompt_text(wrs_transcribe_task* t, const char* utf8);
/* `TranscribeTask::run(&audio)` (transcribe.rs:103-243): the sequential long-form loop. audio [n] f32 host. */
int wrs_transcribe_task_run(wrs_transcribe_task* t, const float* audio, int64_t n_samples, wrs_transcribe_output* out);
/* the same loop from a caller-supplied log-mel spectrogram [80, n_frames] f32 host (run_inner after its
 * `log_mel_spectrogram` call, transcribe.rs:113-243): for callers that keep spectrograms, and for parity tests on the
 * reference's own spectrogram fixture (test/data/mel-spectrogram.csv). */
int wrs_transcribe_task_run_mel(wrs_transcribe_task* t, const float* mel, int64_t n_frames, wrs_transcribe_output* out);
/* fixed-stride variant (= the reference's `seek += N_FRAMES` branch with conditioning off,
 * transcribe.rs:231): windows are independent, decoded `batch` at a time; this is what shards over GPUs. */
int wrs_transcribe_task_run_chunked(wrs_transcribe_task* t, const float* audio, int64_t n_samples, int32_t batch,
                                    wrs_transcribe_output* out);
void wrs_transcribe_output_free(wrs_transcribe_output* o);
void wrs_transcribe_task_destroy(wrs_transcribe_task* t);

/* ---------------------------------------------------------------- measurement hooks */

/* per-kernel device timing of the last wrs_decode_task_run* call, measured with HIP events on the
 * library's own stream.  names[i] is a static string. Returns the number of entries written. */
typedef struct wrs_stage_time {
    const char* name;
    double ms;          /* summed device time */
    int64_t launches;
    double bytes;       /* algorithmic bytes moved, summed over launches (0 if compute-bound) */
    double flops;       /* algorithmic flops, summed over launches */
} wrs_stage_time;
int wrs_decode_task_set_profiling(wrs_decode_task* t, int enabled);
/* host wall-clock split of the last wrs_decode_task_run* call (ms): mel+encoder+cross-K/V (to the first sync),
 * prompt prefill, device decode steps (launch + wait), host sequencer (beam search, parameter upload prep). */
int wrs_decode_task_last_run_times(wrs_decode_task* t, double* encoder_ms, double* prefill_ms, double* steps_ms, double* host_ms);
/* live timing of the dominant decode kernel (cross-attention, HBM-bound) with HIP events on the library's
 * stream during normal runs: every 16th decode step is launched eagerly instead of from the captured graph,
 * with an event pair around the kernel in every `layer_stride`-th layer.  get: summed ms, launches and
 * algorithmic bytes (n_audio * 2 * 1500 * D * 2 B per launch) since the last set. */
int wrs_decode_task_set_cross_timing(wrs_decode_task* t, int enabled, int layer_stride);
int wrs_decode_task_get_cross_timing(wrs_decode_task* t, double* ms, int64_t* launches, double* bytes);
int wrs_decode_task_stage_times(wrs_decode_task* t, wrs_stage_time* out, int32_t cap, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* WRS_H */
