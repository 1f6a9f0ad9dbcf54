// Opaque handle types of include/wrs.h (shared by the translation units that implement the C entry points).
#pragma once
#include "transcribe.h"

struct wrs_model { wrs::Model m; };
struct wrs_decode_task { wrs::DecodeTask t; };
struct wrs_transcribe_task { wrs::TranscribeTask t; };
struct wrs_tokenizer { wrs::Tokenizer t; };
