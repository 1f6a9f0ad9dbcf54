"""ctypes binding of libwrs_mi355x.so (include/wrs.h).  No fallback: if the HIP library is not
built, importing the product API raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(_HERE, "libwrs_mi355x.so")
DEBUG_LIB_PATH = os.path.join(_HERE, "libwrs_mi355x_debug.so")
# WRS_DEBUG_LIB=1 (set by tests/conftest.py and __graft_entry__.smoke() before this module is imported) selects the test build:
# the product's objects + the include/wrs_debug.h hooks, with the experiment switches honoured.  One library per process.
IS_DEBUG_LIB = os.environ.get("WRS_DEBUG_LIB", "0") == "1"
LIB_PATH = DEBUG_LIB_PATH if IS_DEBUG_LIB else PRODUCT_LIB_PATH

WRS_OK, WRS_EINVAL, WRS_ENODEVICE, WRS_EHIP, WRS_ESTATE, WRS_EUNSUPPORTED = 0, 1, 2, 3, 4, 5
WRS_F32, WRS_F16 = 0, 1
WRS_MEM_HOST, WRS_MEM_DEVICE = 0, 1
WRS_EXTRACT_GREEDY, WRS_EXTRACT_BEAM_SEARCH = 0, 1


class WrsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wrs error {code}: {msg}")
        self.code = code


class ModelDimsC(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_vocab", "n_audio_ctxs", "n_audio_states", "n_audio_heads", "n_audio_layers",
                                         "n_text_ctxs", "n_text_states", "n_text_heads", "n_text_layers")]


class TokenIdsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("eot", "sot", "translate", "transcribe", "startoflm", "startofprev", "no_speech",
                                         "no_timestamps", "ts_begin", "space")]


class DecodeOptionsC(C.Structure):
    _fields_ = [("sample_len", C.c_int32), ("extract_mode", C.c_int32), ("group_size", C.c_int32), ("patience", C.c_float),
                ("has_len_penalty", C.c_int32), ("len_penalty", C.c_double), ("has_max_initial_timestamp", C.c_int32),
                ("max_initial_timestamp", C.c_double), ("timestamps", C.c_int32), ("suppress_blank", C.c_int32),
                ("suppress_non_speech", C.c_int32), ("suppress_tokens", C.POINTER(C.c_int32)), ("n_suppress_tokens", C.c_int32),
                ("non_speech_tokens", C.POINTER(C.c_int32)), ("n_non_speech_tokens", C.c_int32),
                ("position_offset_quirk", C.c_int32), ("max_audio", C.c_int32),
                ("sot_sequence_tail", C.POINTER(C.c_int32)), ("n_sot_sequence_tail", C.c_int32)]


class DecodeOutputC(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_int64)), ("n_tokens", C.c_int32), ("avg_logprob", C.c_double),
                ("no_speech_prob", C.c_double), ("sum_logprob", C.c_double), ("audio_features_device", C.c_void_p),
                ("text", C.c_void_p)]


class DecodeResultC(C.Structure):
    _fields_ = [("outputs", C.POINTER(DecodeOutputC)), ("n_outputs", C.c_int32), ("n_steps", C.c_int32)]


class TranscribeOptionsC(C.Structure):
    _fields_ = [("decode", DecodeOptionsC), ("prompt_tokens", C.POINTER(C.c_int64)), ("n_prompt_tokens", C.c_int32),
                ("condition_on_prev_text", C.c_int32)]


class SegmentC(C.Structure):
    _fields_ = [("seek", C.c_int64), ("start_time", C.c_double), ("end_time", C.c_double), ("start_token", C.c_int64),
                ("end_token", C.c_int64), ("text", C.c_void_p)]


class TranscribeOutputC(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_int64)), ("n_tokens", C.c_int64), ("segments", C.POINTER(SegmentC)),
                ("n_segments", C.c_int64), ("window_seek", C.POINTER(C.c_int64)), ("window_avg_logprob", C.POINTER(C.c_double)),
                ("window_no_speech_prob", C.POINTER(C.c_double)), ("n_windows", C.c_int64), ("text", C.c_void_p),
                ("truncated", C.c_int32)]


class StageTimeC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_int64), ("bytes", C.c_double), ("flops", C.c_double)]


# every symbol include/wrs.h declares: name -> (restype, argtypes)
P = C.POINTER
SYMBOLS = {
    "wrs_last_error": (C.c_char_p, []),
    "wrs_version": (C.c_char_p, []),
    "wrs_device_count": (C.c_int, []),
    "wrs_model_create": (C.c_int, [P(ModelDimsC), C.c_int, P(C.c_void_p)]),
    "wrs_model_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, P(C.c_int64), C.c_int, C.c_void_p, C.c_int]),
    "wrs_model_load_safetensors": (C.c_int, [C.c_void_p, C.c_char_p, P(C.c_int32)]),
    "wrs_model_finalize": (C.c_int, [C.c_void_p]),
    "wrs_model_device": (C.c_int, [C.c_void_p]),
    "wrs_model_get_dims": (C.c_int, [C.c_void_p, P(ModelDimsC)]),
    "wrs_model_destroy": (None, [C.c_void_p]),
    "wrs_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "wrs_decoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "wrs_model_update_cache": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "wrs_model_clear_cache": (C.c_int, [C.c_void_p]),
    "wrs_load_audio": (C.c_int, [C.c_char_p, P(P(C.c_float)), P(C.c_int64)]),
    "wrs_resample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, P(P(C.c_float)), P(C.c_int64)]),
    "wrs_audio_free": (None, [P(C.c_float)]),
    "wrs_mel_filter": (C.c_int, [C.c_void_p]),
    "wrs_pad_or_trim": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "wrs_log_mel_spectrogram": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]),
    "wrs_tokenizer_create_from_bytes": (C.c_int, [C.c_void_p, C.c_size_t, P(C.c_void_p)]),
    "wrs_tokenizer_create_from_file": (C.c_int, [C.c_char_p, P(C.c_void_p)]),
    "wrs_tokenizer_destroy": (None, [C.c_void_p]),
    "wrs_tokenizer_token_ids": (C.c_int, [C.c_void_p, P(TokenIdsC)]),
    "wrs_tokenizer_encode": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_int32, P(C.c_int32)]),
    "wrs_tokenizer_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, P(C.c_void_p), P(C.c_size_t)]),
    "wrs_string_free": (None, [C.c_void_p]),
    "wrs_tokenizer_non_speech_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, P(C.c_int32)]),
    "wrs_tokenizer_token_to_id": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "wrs_tokenizer_language_token": (C.c_int32, [C.c_void_p, C.c_char_p]),
    "wrs_tokenizer_vocab_size": (C.c_int32, [C.c_void_p]),
    "wrs_decode_task_set_tokenizer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wrs_transcribe_task_set_tokenizer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wrs_transcribe_task_set_prompt_text": (C.c_int, [C.c_void_p, C.c_char_p]),
    "wrs_decode_task_create": (C.c_int, [C.c_void_p, P(TokenIdsC), P(DecodeOptionsC), P(C.c_void_p)]),
    "wrs_decode_task_set_prompt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "wrs_decode_task_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, P(DecodeResultC)]),
    "wrs_decode_task_run_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, P(DecodeResultC)]),
    "wrs_decode_task_read_audio_features": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "wrs_decode_task_run_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, P(DecodeResultC)]),
    "wrs_decode_task_encode_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int]),
    "wrs_decode_task_run_pending": (C.c_int, [C.c_void_p, P(DecodeResultC)]),
    "wrs_decode_task_run_encoded": (C.c_int, [C.c_void_p, C.c_int32, P(DecodeResultC)]),
    "wrs_decode_task_detect_language": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "wrs_decode_task_set_languages": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "wrs_decode_task_trace": (C.c_int, [C.c_void_p, P(C.c_int32), P(C.c_int32), P(P(C.c_int32)), P(P(C.c_int32)), P(P(C.c_float))]),
    "wrs_decode_result_free": (None, [P(DecodeResultC)]),
    "wrs_decode_task_destroy": (None, [C.c_void_p]),
    "wrs_filter_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "wrs_window_pool_create": (C.c_int, [P(C.c_void_p), C.c_int32, P(TokenIdsC), P(DecodeOptionsC), C.c_int, P(C.c_void_p)]),
    "wrs_window_pool_set_tokenizer": (C.c_int, [C.c_void_p, C.c_void_p]),
    "wrs_window_pool_run_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, P(DecodeResultC)]),
    "wrs_window_pool_run_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, P(DecodeResultC)]),
    "wrs_window_pool_destroy": (None, [C.c_void_p]),
    "wrs_transcribe_task_create": (C.c_int, [C.c_void_p, P(TokenIdsC), P(TranscribeOptionsC), P(C.c_void_p)]),
    "wrs_transcribe_task_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, P(TranscribeOutputC)]),
    "wrs_transcribe_task_run_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, P(TranscribeOutputC)]),
    "wrs_transcribe_task_run_chunked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, P(TranscribeOutputC)]),
    "wrs_transcribe_output_free": (None, [P(TranscribeOutputC)]),
    "wrs_transcribe_task_destroy": (None, [C.c_void_p]),
    "wrs_decode_task_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "wrs_decode_task_stage_times": (C.c_int, [C.c_void_p, P(StageTimeC), C.c_int32, P(C.c_int32)]),
    "wrs_decode_task_last_run_times": (C.c_int, [C.c_void_p, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double)]),
    "wrs_decode_task_set_cross_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "wrs_decode_task_get_cross_timing": (C.c_int, [C.c_void_p, P(C.c_double), P(C.c_int64), P(C.c_double)]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                          "whisper_rs_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != WRS_OK:
        raise WrsError(code, load().wrs_last_error().decode("utf-8", "replace"))
