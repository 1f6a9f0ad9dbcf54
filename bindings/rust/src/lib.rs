//! The reference crate's public surface (laptou/whisper-rs, `src/lib.rs`) re-implemented over the C-ABI of
//! `libwrs_mi355x.so` (include/wrs.h).  Signatures follow the reference with `&[f32]` / `Vec<i64>` where it has
//! `tch::Tensor`; every function cites the reference item it replaces.  Types are `!Send + !Sync` like the reference's
//! (`Rc`/`RefCell`, decode/mod.rs:30-32): raw pointers make them so.
//!
//! NOT compiled in the build image (no Rust toolchain); kept honest by tests/test_rust_ffi_cpu.py, which checks that
//! every `ffi::wrs_*` call here exists in the generated binding with the argument count used.
pub mod ffi;

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};
use std::path::Path;
use std::rc::Rc;

fn check(code: c_int) -> anyhow::Result<()> {
    if code == ffi::WRS_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::wrs_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow::anyhow!("wrs error {code}: {msg}"))
}

unsafe fn take_string(p: *mut c_char) -> Option<String> {
    if p.is_null() {
        None
    } else {
        Some(CStr::from_ptr(p).to_string_lossy().into_owned())
    }
}

/// reference `whisper::audio` (src/audio.rs)
pub mod audio {
    use super::*;
    pub const SAMPLE_RATE: i64 = 16000; // audio.rs:12-18
    pub const N_FFT: i64 = 400;
    pub const N_MELS: i64 = 80;
    pub const HOP_LENGTH: i64 = 160;
    pub const CHUNK_LENGTH: i64 = 30;
    pub const N_SAMPLES: i64 = CHUNK_LENGTH * SAMPLE_RATE;
    pub const N_FRAMES: i64 = N_SAMPLES / HOP_LENGTH;

    /// `audio::load_audio(path)` (audio.rs:20-173): mono f32 at 16 kHz (RIFF/WAVE input)
    pub fn load_audio(path: impl AsRef<Path>) -> anyhow::Result<Vec<f32>> {
        let c = CString::new(path.as_ref().to_string_lossy().as_bytes())?;
        let mut p: *mut f32 = std::ptr::null_mut();
        let mut n: i64 = 0;
        check(unsafe { ffi::wrs_load_audio(c.as_ptr(), &mut p, &mut n) })?;
        let v = unsafe { std::slice::from_raw_parts(p, n as usize) }.to_vec();
        unsafe { ffi::wrs_audio_free(p) };
        Ok(v)
    }
    /// `audio::mel_filter()` (audio.rs:175-188): [80, 201] row-major
    pub fn mel_filter() -> Vec<f32> {
        let mut f = vec![0f32; 80 * 201];
        unsafe { ffi::wrs_mel_filter(f.as_mut_ptr()) };
        f
    }
    /// `audio::pad_or_trim(x, len)` (audio.rs:213-228) on the last dim of a [rows, n] array
    pub fn pad_or_trim(x: &[f32], rows: usize, len: usize) -> anyhow::Result<Vec<f32>> {
        anyhow::ensure!(rows > 0 && x.len() % rows == 0, "pad_or_trim: {} values are not {} equal rows", x.len(), rows);
        let n = x.len() / rows;
        let mut out = vec![0f32; rows * len];
        unsafe { ffi::wrs_pad_or_trim(x.as_ptr(), rows as i64, n as i64, len as i64, out.as_mut_ptr()) };
        Ok(out)
    }
    /// `audio::log_mel_spectrogram(audio, mel_filter)` (audio.rs:230-259): [80, n / 160]
    pub fn log_mel_spectrogram(audio: &[f32], mel_filter: Option<&[f32]>, device: i32) -> anyhow::Result<Vec<f32>> {
        anyhow::ensure!(audio.len() >= 160, "log_mel_spectrogram: {} samples are less than one hop", audio.len());
        if let Some(f) = mel_filter {
            anyhow::ensure!(f.len() == 80 * 201, "log_mel_spectrogram: the mel filter must be [80, 201], got {} values", f.len());
        }
        let mut out = vec![0f32; 80 * (audio.len() / 160)];
        let filt = mel_filter.map_or(std::ptr::null(), |f| f.as_ptr());
        check(unsafe { ffi::wrs_log_mel_spectrogram(device, audio.as_ptr(), 1, audio.len() as i64, filt, out.as_mut_ptr(), ffi::WRS_MEM_HOST) })?;
        Ok(out)
    }
}

/// reference `ModelDims` (model.rs:546-557)
pub type ModelDims = ffi::WrsModelDims;

/// reference `Whisper` (model.rs:538-604): owner of the device-side model
pub struct Whisper {
    pub(crate) handle: *mut ffi::WrsModel,
    pub dims: ModelDims,
}

impl Whisper {
    /// `Whisper::new(vs.root(), dims)` + `vars.load(path)` (model.rs:560-584, examples/transcribe.rs:22-37): `weights` is a
    /// safetensors file holding the reference's VarStore tensor names
    pub fn new(weights: impl AsRef<Path>, dims: ModelDims, device: i32) -> anyhow::Result<Self> {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_model_create(&dims, device, &mut handle) })?;
        let m = Whisper { handle, dims };
        let c = CString::new(weights.as_ref().to_string_lossy().as_bytes())?;
        let mut n = 0i32;
        check(unsafe { ffi::wrs_model_load_safetensors(m.handle, c.as_ptr(), &mut n) })?;
        check(unsafe { ffi::wrs_model_finalize(m.handle) })?;
        Ok(m)
    }
    /// `Whisper::device()` (model.rs:601-603)
    pub fn device(&self) -> i32 {
        unsafe { ffi::wrs_model_device(self.handle) }
    }
    /// `AudioEncoder::forward` (model.rs:417-441): mel [n, 80, 3000] -> [n, 1500, D]
    pub fn encoder_forward(&self, mel: &[f32], n_audio: usize) -> anyhow::Result<Vec<f32>> {
        anyhow::ensure!(n_audio > 0 && mel.len() == n_audio * 80 * 3000, "encoder_forward: mel must be [{}, 80, 3000], got {} values", n_audio, mel.len());
        let mut out = vec![0f32; n_audio * (self.dims.n_audio_ctxs * self.dims.n_audio_states) as usize];
        check(unsafe { ffi::wrs_encoder_forward(self.handle, mel.as_ptr(), n_audio as c_int, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `TextDecoder::forward_ext(tokens, xa, offset)` (model.rs:490-522): tokens [b, s] -> logits [b, s, V]
    pub fn decoder_forward_ext(&self, tokens: &[i64], n_batch: usize, xa: Option<&[f32]>, offset: i64) -> anyhow::Result<Vec<f32>> {
        anyhow::ensure!(n_batch > 0 && !tokens.is_empty() && tokens.len() % n_batch == 0, "decoder_forward_ext: {} tokens are not [{}, s]", tokens.len(), n_batch);
        if let Some(x) = xa {
            let per = (self.dims.n_audio_ctxs * self.dims.n_audio_states) as usize;
            anyhow::ensure!(x.len() == n_batch * per, "decoder_forward_ext: xa must be [{}, n_audio_ctxs, n_audio_states], got {} values", n_batch, x.len());
        }
        anyhow::ensure!(offset >= 0, "decoder_forward_ext: negative offset");
        let s = tokens.len() / n_batch;
        let mut out = vec![0f32; n_batch * s * self.dims.n_vocab as usize];
        let xa = xa.map_or(std::ptr::null(), |x| x.as_ptr());
        check(unsafe { ffi::wrs_decoder_forward(self.handle, tokens.as_ptr(), n_batch as c_int, s as c_int, xa, offset, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `Whisper::update_cache(&sources_indices)` (model.rs:591-594)
    pub fn update_cache(&self, src: &[i64]) -> anyhow::Result<()> {
        anyhow::ensure!(!src.is_empty(), "update_cache: empty source index list");
        check(unsafe { ffi::wrs_model_update_cache(self.handle, src.as_ptr(), src.len() as c_int) })
    }
    /// `Whisper::clear_cache()` (model.rs:596-599)
    pub fn clear_cache(&self) -> anyhow::Result<()> {
        check(unsafe { ffi::wrs_model_clear_cache(self.handle) })
    }
}
impl Drop for Whisper {
    fn drop(&mut self) {
        unsafe { ffi::wrs_model_destroy(self.handle) }
    }
}

/// reference `Tokenizer` (tokenize.rs:111-321)
pub struct Tokenizer {
    pub(crate) handle: *mut ffi::WrsTokenizer,
    pub ids: ffi::WrsTokenIds,
}
impl Tokenizer {
    /// `Tokenizer::new_from_file(task, path)` (tokenize.rs:204-207)
    pub fn new_from_file(path: impl AsRef<Path>) -> anyhow::Result<Self> {
        let c = CString::new(path.as_ref().to_string_lossy().as_bytes())?;
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_tokenizer_create_from_file(c.as_ptr(), &mut handle) })?;
        Self::finish(handle)
    }
    /// `Tokenizer::new_from_bytes(task, bytes)` (tokenize.rs:142-202)
    pub fn new_from_bytes(bytes: &[u8]) -> anyhow::Result<Self> {
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_tokenizer_create_from_bytes(bytes.as_ptr() as *const _, bytes.len(), &mut handle) })?;
        Self::finish(handle)
    }
    fn finish(handle: *mut ffi::WrsTokenizer) -> anyhow::Result<Self> {
        let mut ids = unsafe { std::mem::zeroed::<ffi::WrsTokenIds>() };
        check(unsafe { ffi::wrs_tokenizer_token_ids(handle, &mut ids) })?;
        Ok(Tokenizer { handle, ids })
    }
    /// `tokenizer.encode(text, true)` (tokenize.rs:314-320)
    pub fn encode(&self, text: &str) -> anyhow::Result<Vec<u32>> {
        let mut out = vec![0i32; text.len() + 8];
        let mut n = 0i32;
        check(unsafe { ffi::wrs_tokenizer_encode(self.handle, text.as_ptr() as *const c_char, text.len(), out.as_mut_ptr(), out.len() as i32, &mut n) })?;
        out.truncate(n as usize);
        Ok(out.into_iter().map(|t| t as u32).collect())
    }
    /// `Tokenizer::decode(&tokens)` (tokenize.rs:309-312)
    pub fn decode(&self, tokens: &[i64]) -> anyhow::Result<String> {
        let mut p: *mut c_char = std::ptr::null_mut();
        let mut len = 0usize;
        check(unsafe { ffi::wrs_tokenizer_decode(self.handle, tokens.as_ptr(), tokens.len() as i64, &mut p, &mut len) })?;
        let s = String::from_utf8_lossy(unsafe { std::slice::from_raw_parts(p as *const u8, len) }).into_owned();
        unsafe { ffi::wrs_string_free(p) };
        Ok(s)
    }
    /// `Tokenizer::non_speech_tokens()` (tokenize.rs:215-307)
    pub fn non_speech_tokens(&self) -> anyhow::Result<Vec<u32>> {
        let mut out = vec![0i32; 256];
        let mut n = 0i32;
        check(unsafe { ffi::wrs_tokenizer_non_speech_tokens(self.handle, out.as_mut_ptr(), 256, &mut n) })?;
        out.truncate(n as usize);
        Ok(out.into_iter().map(|t| t as u32).collect())
    }
}
impl Drop for Tokenizer {
    fn drop(&mut self) {
        unsafe { ffi::wrs_tokenizer_destroy(self.handle) }
    }
}

/// reference `TokenExtractMode` (decode/extractor/mod.rs:12-16), field spelling as in the reference
#[derive(Debug, Clone, Copy)]
pub enum TokenExtractMode {
    Greedy { group_sze: usize },
    BeamSearch { beam_size: usize, patience: f32 },
}

/// reference `DecodeOptions` (decode/mod.rs:95-106)
#[derive(Debug, Clone)]
pub struct DecodeOptions {
    pub sample_len: Option<usize>,
    pub token_extract_mode: TokenExtractMode,
    pub len_penalty: Option<f64>,
    pub max_initial_timestamp: Option<f64>,
    pub timestamps: bool,
    pub suppress_blank: bool,
    pub suppress_non_speech: bool,
    pub suppress_tokens: Option<Vec<u32>>,
    pub prompt: Option<Vec<i64>>,
}

/// reference `DecodeOutput` (decode/mod.rs:109-119); `audio_features` stays on the device (pointer valid until the model's
/// next encoder pass; `DecodeTask::audio_features` copies it out)
#[derive(Debug, Clone)]
pub struct DecodeOutput {
    pub audio_features_device: *const f32,
    pub tokens: Vec<i64>,
    pub text: String,
    pub avg_logprob: f64,
    pub no_speech_prob: f64,
}

struct OptionBuffers {
    suppress: Vec<i32>,
    non_speech: Vec<i32>,
}

fn fill_options(o: &DecodeOptions, tok: &Tokenizer, max_audio: i32) -> anyhow::Result<(ffi::WrsDecodeOptions, OptionBuffers)> {
    let bufs = OptionBuffers {
        suppress: o.suppress_tokens.clone().unwrap_or_default().into_iter().map(|t| t as i32).collect(),
        non_speech: tok.non_speech_tokens()?.into_iter().map(|t| t as i32).collect(),
    };
    let (mode, group, patience) = match o.token_extract_mode {
        TokenExtractMode::Greedy { group_sze } => (ffi::WRS_EXTRACT_GREEDY, group_sze as i32, 1.0),
        TokenExtractMode::BeamSearch { beam_size, patience } => (ffi::WRS_EXTRACT_BEAM_SEARCH, beam_size as i32, patience),
    };
    let c = ffi::WrsDecodeOptions {
        sample_len: o.sample_len.map_or(0, |s| s as i32),
        extract_mode: mode,
        group_size: group,
        patience,
        has_len_penalty: o.len_penalty.is_some() as i32,
        len_penalty: o.len_penalty.unwrap_or(0.0),
        has_max_initial_timestamp: o.max_initial_timestamp.is_some() as i32,
        max_initial_timestamp: o.max_initial_timestamp.unwrap_or(0.0),
        timestamps: o.timestamps as i32,
        suppress_blank: o.suppress_blank as i32,
        suppress_non_speech: o.suppress_non_speech as i32,
        suppress_tokens: if o.suppress_tokens.is_some() { bufs.suppress.as_ptr() } else { std::ptr::null() },
        n_suppress_tokens: bufs.suppress.len() as i32,
        non_speech_tokens: bufs.non_speech.as_ptr(),
        n_non_speech_tokens: bufs.non_speech.len() as i32,
        position_offset_quirk: 1, // the reference's decoder offset = loop index (decode/mod.rs:244)
        max_audio,
        sot_sequence_tail: std::ptr::null(), // Tokenizer::sequence_sot() == [sot] (tokenize.rs:209-213)
        n_sot_sequence_tail: 0,
    };
    Ok((c, bufs))
}

/// reference `DecodeTask` (decode/mod.rs:24-419)
pub struct DecodeTask<'a> {
    _model: &'a Whisper,
    tokenizer: Rc<Tokenizer>,
    handle: *mut ffi::WrsDecodeTask,
}
impl<'a> DecodeTask<'a> {
    /// `DecodeTask::new(&model, tokenizer, options)` (decode/mod.rs:122-204)
    pub fn new(model: &'a Whisper, tokenizer: Rc<Tokenizer>, options: DecodeOptions) -> anyhow::Result<Self> {
        let (c, _bufs) = fill_options(&options, &tokenizer, 1)?;
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_decode_task_create(model.handle, &tokenizer.ids, &c, &mut handle) })?; // option arrays are copied by the call
        check(unsafe { ffi::wrs_decode_task_set_tokenizer(handle, tokenizer.handle) })?;
        let mut t = DecodeTask { _model: model, tokenizer, handle };
        t.set_prompt(options.prompt.as_deref());
        Ok(t)
    }
    /// `DecodeTask::set_prompt(Option<&Tensor>)` (decode/mod.rs:206-218)
    pub fn set_prompt(&mut self, prompt: Option<&[i64]>) {
        let (p, n) = prompt.map_or((std::ptr::null(), 0), |p| (p.as_ptr(), p.len() as i32));
        unsafe { ffi::wrs_decode_task_set_prompt(self.handle, p, n) };
    }
    /// `DecodeTask::run(&mel)` (decode/mod.rs:284-419): mel [80, 3000]
    pub fn run(&mut self, mel: &[f32]) -> anyhow::Result<Vec<DecodeOutput>> {
        anyhow::ensure!(mel.len() == 80 * 3000, "DecodeTask::run: mel must be [80, 3000], got {} values", mel.len());
        let mut r = ffi::WrsDecodeResult { outputs: std::ptr::null_mut(), n_outputs: 0, n_steps: 0 };
        check(unsafe { ffi::wrs_decode_task_run(self.handle, mel.as_ptr(), 1, ffi::WRS_MEM_HOST, &mut r) })?;
        let outs = collect_outputs(&mut r);
        let _ = &self.tokenizer;
        Ok(outs)
    }
    /// host copy of `DecodeOutput.audio_features` (decode/mod.rs:110) of window `index` of the batch decoded last:
    /// [n_audio_ctxs * n_audio_states] f32.  Fails once the model has run its encoder again.
    pub fn audio_features(&mut self, index: usize, dims: &ModelDims) -> anyhow::Result<Vec<f32>> {
        let mut out = vec![0f32; (dims.n_audio_ctxs * dims.n_audio_states) as usize];
        check(unsafe { ffi::wrs_decode_task_read_audio_features(self.handle, index as i32, out.as_mut_ptr(), out.len() as i64) })?;
        Ok(out)
    }
}
fn collect_outputs(r: &mut ffi::WrsDecodeResult) -> Vec<DecodeOutput> {
    let outs = unsafe { std::slice::from_raw_parts(r.outputs, r.n_outputs as usize) }
        .iter()
        .map(|o| DecodeOutput {
            audio_features_device: o.audio_features_device,
            tokens: unsafe { std::slice::from_raw_parts(o.tokens, o.n_tokens as usize) }.to_vec(),
            text: unsafe { take_string(o.text) }.unwrap_or_default(),
            avg_logprob: o.avg_logprob,
            no_speech_prob: o.no_speech_prob,
        })
        .collect();
    unsafe { ffi::wrs_decode_result_free(r) };
    outs
}

/// Beyond the reference (whose loop is sequential on one device): independent 30-s windows sharded over model replicas —
/// one replica per GPU of the node, or several per GPU — each with its own worker thread and decode task pulling batches of
/// `windows_per_batch` windows from a shared queue inside the library (no collective).  Outputs come back in window order.
pub struct WindowPool<'a> {
    _models: Vec<&'a Whisper>,
    _tokenizer: Rc<Tokenizer>,
    handle: *mut ffi::WrsWindowPool,
}
impl<'a> WindowPool<'a> {
    /// `models`: one `Whisper` per worker (e.g. `Whisper::new(path, dims, gpu)` for gpu in 0..8); `options` as for `DecodeTask`
    pub fn new(models: &[&'a Whisper], tokenizer: Rc<Tokenizer>, options: DecodeOptions, windows_per_batch: usize, overlapped: bool) -> anyhow::Result<Self> {
        let (c, _bufs) = fill_options(&options, &tokenizer, windows_per_batch as i32)?;
        let handles: Vec<*mut ffi::WrsModel> = models.iter().map(|m| m.handle).collect();
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_window_pool_create(handles.as_ptr(), handles.len() as i32, &tokenizer.ids, &c, overlapped as i32, &mut handle) })?;
        check(unsafe { ffi::wrs_window_pool_set_tokenizer(handle, tokenizer.handle) })?;
        Ok(WindowPool { _models: models.to_vec(), _tokenizer: tokenizer, handle })
    }
    /// `audio`: n_windows x 480 000 samples, every window its own recording (its own log-mel normalisation)
    pub fn run_audio(&mut self, audio: &[f32]) -> anyhow::Result<Vec<DecodeOutput>> {
        anyhow::ensure!(audio.len() % 480_000 == 0 && !audio.is_empty(), "audio must be a whole number of 30-s windows");
        let mut r = ffi::WrsDecodeResult { outputs: std::ptr::null_mut(), n_outputs: 0, n_steps: 0 };
        check(unsafe { ffi::wrs_window_pool_run_audio(self.handle, audio.as_ptr(), (audio.len() / 480_000) as i64, ffi::WRS_MEM_HOST, &mut r) })?;
        Ok(collect_outputs(&mut r))
    }
    /// the windows of ONE recording's spectrogram [80, n_frames] (normalised over the whole file as `transcribe.rs:111-112` does),
    /// cut at the fixed 3000-frame stride
    pub fn run_mel(&mut self, mel: &[f32], n_frames: usize) -> anyhow::Result<Vec<DecodeOutput>> {
        anyhow::ensure!(mel.len() == 80 * n_frames && n_frames > 0, "mel must be [80, n_frames]");
        let mut r = ffi::WrsDecodeResult { outputs: std::ptr::null_mut(), n_outputs: 0, n_steps: 0 };
        check(unsafe { ffi::wrs_window_pool_run_mel(self.handle, mel.as_ptr(), n_frames as i64, &mut r) })?;
        Ok(collect_outputs(&mut r))
    }
}
impl Drop for WindowPool<'_> {
    fn drop(&mut self) {
        unsafe { ffi::wrs_window_pool_destroy(self.handle) }
    }
}

impl Drop for DecodeTask<'_> {
    fn drop(&mut self) {
        unsafe { ffi::wrs_decode_task_destroy(self.handle) }
    }
}

/// reference `TranscribePrompt` (transcribe.rs:25-29)
#[derive(Debug, Clone)]
pub enum TranscribePrompt {
    Pretokenized(Vec<u32>),
    Text(String),
    None { condition_on_prev_text: bool },
}

/// reference `TranscribeOptions` (transcribe.rs:32-50)
pub struct TranscribeOptions {
    pub decode: DecodeOptions,
    pub prompt: TranscribePrompt,
    pub tokenizer: Rc<Tokenizer>,
}

/// reference `TranscribeOutputSegment` (transcribe.rs:59-67)
#[derive(Debug, Clone)]
pub struct TranscribeOutputSegment {
    pub seek: i64,
    pub start_time: f64,
    pub end_time: f64,
    pub start_token: i64,
    pub end_token: i64,
    pub text: String,
}

/// reference `TranscribeOutput` (transcribe.rs:53-57) (+ `truncated`: the loop stopped before the end of the audio)
#[derive(Debug, Clone)]
pub struct TranscribeOutput {
    pub tokens: Vec<i64>,
    pub text: String,
    pub segments: Vec<TranscribeOutputSegment>,
    pub truncated: bool,
}

/// reference `TranscribeTask` (transcribe.rs:13-244)
pub struct TranscribeTask<'a> {
    _model: &'a Whisper,
    _tokenizer: Rc<Tokenizer>,
    handle: *mut ffi::WrsTranscribeTask,
}
impl<'a> TranscribeTask<'a> {
    /// `TranscribeTask::new(&model, options)` (transcribe.rs:70-101)
    pub fn new(model: &'a Whisper, options: TranscribeOptions) -> anyhow::Result<Self> {
        let (decode, _bufs) = fill_options(&options.decode, &options.tokenizer, 1)?;
        let pre: Vec<i64> = match &options.prompt {
            TranscribePrompt::Pretokenized(v) => v.iter().map(|&t| t as i64).collect(),
            _ => Vec::new(),
        };
        let c = ffi::WrsTranscribeOptions {
            decode,
            prompt_tokens: if matches!(options.prompt, TranscribePrompt::Pretokenized(_)) { pre.as_ptr() } else { std::ptr::null() },
            n_prompt_tokens: pre.len() as i32,
            condition_on_prev_text: match options.prompt {
                TranscribePrompt::None { condition_on_prev_text } => condition_on_prev_text as i32,
                _ => 1, // any explicit prompt implies conditioning (transcribe.rs:142-147)
            },
        };
        let mut handle = std::ptr::null_mut();
        check(unsafe { ffi::wrs_transcribe_task_create(model.handle, &options.tokenizer.ids, &c, &mut handle) })?;
        check(unsafe { ffi::wrs_transcribe_task_set_tokenizer(handle, options.tokenizer.handle) })?;
        if let TranscribePrompt::Text(t) = &options.prompt {
            let ct = CString::new(t.as_str())?;
            check(unsafe { ffi::wrs_transcribe_task_set_prompt_text(handle, ct.as_ptr()) })?; // transcribe.rs:130-134
        }
        Ok(TranscribeTask { _model: model, _tokenizer: options.tokenizer, handle })
    }
    /// `TranscribeTask::run(&audio)` (transcribe.rs:103-243): mono 16 kHz samples
    pub fn run(&mut self, audio: &[f32]) -> anyhow::Result<TranscribeOutput> {
        let mut o = unsafe { std::mem::zeroed::<ffi::WrsTranscribeOutput>() };
        check(unsafe { ffi::wrs_transcribe_task_run(self.handle, audio.as_ptr(), audio.len() as i64, &mut o) })?;
        let segments = unsafe { std::slice::from_raw_parts(o.segments, o.n_segments as usize) }
            .iter()
            .map(|s| TranscribeOutputSegment {
                seek: s.seek,
                start_time: s.start_time,
                end_time: s.end_time,
                start_token: s.start_token,
                end_token: s.end_token,
                text: unsafe { take_string(s.text) }.unwrap_or_default(),
            })
            .collect();
        let out = TranscribeOutput {
            tokens: unsafe { std::slice::from_raw_parts(o.tokens, o.n_tokens as usize) }.to_vec(),
            text: unsafe { take_string(o.text) }.unwrap_or_default(),
            segments,
            truncated: o.truncated != 0,
        };
        unsafe { ffi::wrs_transcribe_output_free(&mut o) };
        Ok(out)
    }
}
impl Drop for TranscribeTask<'_> {
    fn drop(&mut self) {
        unsafe { ffi::wrs_transcribe_task_destroy(self.handle) }
    }
}
