// GPT-2 byte-level BPE tokenizer + Whisper special ids: the reference's `Tokenizer`
// (src/tokenize.rs:111-321), which wraps `tokenizers::Tokenizer` 0.13.2 loaded from
// tokenizer/gpt2.json.  Host-only (no device work): SURVEY.md section 8(f) rank 1.
#pragma once
#include "common.h"
#include <map>
#include <unordered_map>

namespace wrs {

struct Tokenizer {
    // --- `tokenizers::Tokenizer` state (what the JSON file holds + added tokens)
    std::vector<std::string> id_to_tok;                       // model vocab + added tokens, "" where unused
    std::unordered_map<std::string, int32_t> tok_to_id;       // model vocab (byte-level strings)
    std::unordered_map<uint64_t, std::pair<int32_t, int32_t>> merges;  // (left id << 32 | right id) -> (rank, merged id)
    struct Added { std::string content; int32_t id; bool special; };
    std::vector<Added> added;                                 // in insertion order
    std::unordered_map<std::string, int32_t> added_to_id;
    bool multilingual = false;

    // --- reference `Tokenizer` fields (tokenize.rs:113-131)
    wrs_token_ids ids{};
    int32_t first_language_id = 0, n_languages = 0;

    // `Tokenizer::new_from_bytes(task, bytes)` (tokenize.rs:142-202).  multilingual = OpenAI's layout for the
    // multilingual vocabulary (one more base token, so every special id is +1); the reference itself only
    // implements the English one.
    void load_json(const char* bytes, size_t n, bool multilingual_layout);
    int32_t add_special_token(const std::string& content);    // `add_special_tokens` (tokenize.rs:168)

    // `tokenizers::Tokenizer::encode(text, add_special_tokens)` through `Deref` (tokenize.rs:314-320)
    std::vector<int32_t> encode(const std::string& utf8) const;
    // `Tokenizer::decode(&tokens)` (tokenize.rs:309-312): skip_special_tokens = false, unknown ids dropped
    std::string decode(const int64_t* ids, int64_t n) const;
    // `Tokenizer::non_speech_tokens()` (tokenize.rs:215-307)
    std::vector<int32_t> non_speech_tokens() const;
    // `Tokenizer::sequence_sot()` (tokenize.rs:209-213); language/task >= 0 give OpenAI's [sot, lang, task]
    std::vector<int32_t> sequence_sot(int32_t language_id, int32_t task_id) const;
    int32_t token_to_id(const std::string& tok) const;        // -1 if absent
    int32_t language_token(const std::string& code) const;    // "<|xx|>" id or -1

  private:
    void bpe_word(const std::string& piece_bytes, std::vector<int32_t>& out) const;
    void encode_plain(const std::string& text, std::vector<int32_t>& out) const;
    mutable std::unordered_map<std::string, std::vector<int32_t>> cache;
};

extern const char* const kLanguageCodes[99];

}  // namespace wrs
