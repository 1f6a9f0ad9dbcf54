// GPT-2 byte-level BPE (the algorithm of `tokenizers` 0.13.2 for a tokenizer.json with model BPE,
// pre_tokenizer/decoder ByteLevel(add_prefix_space=false, use_regex=true), no normalizer) and the reference's
// special-token bookkeeping (src/tokenize.rs:142-312).  `tokenizers` is a third-party dependency of the reference
// (Cargo.lock:1296) that is not vendored; parity is pinned by tests/test_tokenizer_cpu.py against the `tokenizers`
// Python package (same Rust core) on the reference's own tokenizer/gpt2.json.
#include "tokenizer.h"
#include "json_min.h"
#include <algorithm>
#include <cmath>

namespace wrs {

#include "unicode_tables.inc"

const char* const kLanguageCodes[99] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi", "vi",
    "iw", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml", "cy", "sk",
    "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw",
    "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo",
    "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su"};

namespace {
using json::JVal;
using json::append_utf8;

// ------------------------------------------------------------------------------------------ unicode helpers
bool in_ranges(const uint32_t (*r)[2], int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid][0]) hi = mid - 1;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
bool is_letter(uint32_t cp) { return in_ranges(kUnicodeL, kUnicodeL_n, cp); }
bool is_number(uint32_t cp) { return in_ranges(kUnicodeN, kUnicodeN_n, cp); }
// `\s` of the pattern: Unicode White_Space
bool is_space(uint32_t cp) {
    return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}

// Decodes UTF-8 the way Rust's String::from_utf8_lossy does: each maximal ill-formed subpart becomes one U+FFFD.
// Returns code points with the byte offset at which each starts (offsets.size() == cps.size() + 1).
void decode_utf8_lossy(const std::string& s, std::vector<uint32_t>& cps, std::vector<uint32_t>* offsets) {
    const size_t n = s.size();
    size_t i = 0;
    while (i < n) {
        if (offsets) offsets->push_back((uint32_t)i);
        const uint8_t b0 = (uint8_t)s[i];
        if (b0 < 0x80) { cps.push_back(b0); ++i; continue; }
        int need;
        uint32_t cp;
        uint8_t lo = 0x80, hi = 0xBF;
        if (b0 >= 0xC2 && b0 <= 0xDF) { need = 1; cp = b0 & 0x1F; }
        else if (b0 >= 0xE0 && b0 <= 0xEF) { need = 2; cp = b0 & 0x0F; if (b0 == 0xE0) lo = 0xA0; if (b0 == 0xED) hi = 0x9F; }
        else if (b0 >= 0xF0 && b0 <= 0xF4) { need = 3; cp = b0 & 0x07; if (b0 == 0xF0) lo = 0x90; if (b0 == 0xF4) hi = 0x8F; }
        else { cps.push_back(0xFFFD); ++i; continue; }
        size_t j = i + 1;
        bool ok = true;
        for (int k = 0; k < need; ++k) {
            if (j >= n) { ok = false; break; }
            const uint8_t b = (uint8_t)s[j];
            if (b < lo || b > hi) { ok = false; break; }
            cp = (cp << 6) | (b & 0x3F);
            lo = 0x80; hi = 0xBF;
            ++j;
        }
        cps.push_back(ok ? cp : 0xFFFD);
        i = j;
    }
    if (offsets) offsets->push_back((uint32_t)n);
}

// GPT-2 bytes_to_unicode: printable bytes map to themselves, the other 68 to U+0100.. in byte order
struct ByteMap {
    uint32_t byte_to_cp[256];
    std::unordered_map<uint32_t, uint8_t> cp_to_byte;
    std::string byte_to_str[256];
    ByteMap() {
        bool direct[256] = {false};
        for (int b = '!'; b <= '~'; ++b) direct[b] = true;
        for (int b = 0xA1; b <= 0xAC; ++b) direct[b] = true;
        for (int b = 0xAE; b <= 0xFF; ++b) direct[b] = true;
        uint32_t next = 256;
        for (int b = 0; b < 256; ++b) {
            byte_to_cp[b] = direct[b] ? (uint32_t)b : next++;
            cp_to_byte[byte_to_cp[b]] = (uint8_t)b;
            append_utf8(byte_to_str[b], byte_to_cp[b]);
        }
    }
};
const ByteMap& byte_map() {
    static const ByteMap m;
    return m;
}
}  // namespace

// ------------------------------------------------------------------------------------------ loading
int32_t Tokenizer::add_special_token(const std::string& content) {
    auto it = added_to_id.find(content);
    if (it != added_to_id.end()) return it->second;
    auto iv = tok_to_id.find(content);
    int32_t id;
    if (iv != tok_to_id.end()) id = iv->second;  // `add_tokens`: a token the model already has keeps its id
    else {
        id = (int32_t)id_to_tok.size();
        id_to_tok.push_back(content);
    }
    added.push_back({content, id, true});
    added_to_id[content] = id;
    return id;
}

void Tokenizer::load_json(const char* bytes, size_t n, bool multilingual_layout) {
    WRS_REQUIRE(bytes && n > 0, "tokenizer: empty input");
    const JVal root = json::parse(bytes, n);
    WRS_REQUIRE(root.kind == JVal::Obj, "tokenizer json: top level is not an object");
    const JVal* model = root.get("model");
    WRS_REQUIRE(model && model->kind == JVal::Obj, "tokenizer json: no \"model\"");
    const JVal* type = model->get("type");
    WRS_REQUIRE(!type || type->kind != JVal::Str || type->s == "BPE", "tokenizer json: model type must be BPE");
    const JVal* vocab = model->get("vocab");
    const JVal* mg = model->get("merges");
    WRS_REQUIRE(vocab && vocab->kind == JVal::Obj && mg && mg->kind == JVal::Arr, "tokenizer json: BPE model needs vocab and merges");
    if (const JVal* pre = root.get("pre_tokenizer"))
        if (pre->kind == JVal::Obj) {
            const JVal* t = pre->get("type");
            WRS_REQUIRE(t && t->kind == JVal::Str && t->s == "ByteLevel", "tokenizer json: only the ByteLevel pre-tokenizer is supported");
            const JVal* aps = pre->get("add_prefix_space");
            WRS_REQUIRE(!aps || aps->kind != JVal::Bool || !aps->b, "tokenizer json: add_prefix_space=true is not supported");
        }
    if (const JVal* norm = root.get("normalizer")) WRS_REQUIRE(norm->kind == JVal::Null, "tokenizer json: normalizers are not supported");

    int64_t max_id = -1;
    for (auto& kv : vocab->o) {
        WRS_REQUIRE(kv.second.kind == JVal::Num && kv.second.num >= 0 && kv.second.num < (1 << 24), "tokenizer json: bad vocab id");
        max_id = std::max<int64_t>(max_id, (int64_t)kv.second.num);
    }
    id_to_tok.assign((size_t)(max_id + 1), std::string());
    tok_to_id.reserve(vocab->o.size() * 2);
    for (auto& kv : vocab->o) {
        const int32_t id = (int32_t)kv.second.num;
        id_to_tok[id] = kv.first;
        tok_to_id[kv.first] = id;
    }
    merges.reserve(mg->a.size() * 2);
    int32_t rank = 0;
    for (auto& m : mg->a) {
        std::string l, r;
        if (m.kind == JVal::Str) {
            const size_t sp = m.s.find(' ');
            WRS_REQUIRE(sp != std::string::npos, "tokenizer json: bad merge entry");
            l = m.s.substr(0, sp);
            r = m.s.substr(sp + 1);
        } else {
            WRS_REQUIRE(m.kind == JVal::Arr && m.a.size() == 2 && m.a[0].kind == JVal::Str && m.a[1].kind == JVal::Str, "tokenizer json: bad merge entry");
            l = m.a[0].s;
            r = m.a[1].s;
        }
        auto il = tok_to_id.find(l), ir = tok_to_id.find(r), im = tok_to_id.find(l + r);
        WRS_REQUIRE(il != tok_to_id.end() && ir != tok_to_id.end() && im != tok_to_id.end(), "tokenizer json: merge refers to a token missing from the vocab");
        merges.emplace(((uint64_t)(uint32_t)il->second << 32) | (uint32_t)ir->second, std::make_pair(rank, im->second));
        ++rank;
    }
    if (const JVal* at = root.get("added_tokens"))
        if (at->kind == JVal::Arr)
            for (auto& t : at->a) {
                const JVal* id = t.get("id");
                const JVal* content = t.get("content");
                const JVal* special = t.get("special");
                WRS_REQUIRE(id && id->kind == JVal::Num && content && content->kind == JVal::Str, "tokenizer json: bad added token");
                const int32_t i = (int32_t)id->num;
                if ((size_t)i >= id_to_tok.size()) id_to_tok.resize(i + 1);
                if (id_to_tok[i].empty()) id_to_tok[i] = content->s;
                added.push_back({content->s, i, special && special->kind == JVal::Bool && special->b});
                added_to_id[content->s] = i;
            }
    multilingual = multilingual_layout;

    // special tokens in the reference's order (tokenize.rs:154-168); language keys in the declaration order of its
    // table (the reference iterates a HashMap there, i.e. an unspecified order — SURVEY.md quirk list)
    ids.sot = add_special_token("<|startoftranscript|>");
    first_language_id = -1;
    n_languages = 99;
    for (int i = 0; i < 99; ++i) {
        const int32_t id = add_special_token(std::string("<|") + kLanguageCodes[i] + "|>");
        if (i == 0) first_language_id = id;
    }
    ids.translate = add_special_token("<|translate|>");
    ids.transcribe = add_special_token("<|transcribe|>");
    ids.startoflm = add_special_token("<|startoflm|>");
    ids.startofprev = add_special_token("<|startofprev|>");
    ids.no_speech = add_special_token("<|nospeech|>");
    ids.no_timestamps = add_special_token("<|notimestamps|>");
    const int32_t eot = token_to_id("<|endoftext|>");
    WRS_REQUIRE(eot >= 0, "tokenizer: vocabulary has no <|endoftext|>");
    ids.eot = eot;
    ids.ts_begin = ids.no_timestamps + 1;  // tokenize.rs:197: timestamps are the ids past the last special token
    const std::vector<int32_t> sp = encode(" ");
    WRS_REQUIRE(!sp.empty(), "tokenizer: cannot encode a space");
    ids.space = sp[0];
}

int32_t Tokenizer::token_to_id(const std::string& tok) const {
    auto ia = added_to_id.find(tok);
    if (ia != added_to_id.end()) return ia->second;
    auto it = tok_to_id.find(tok);
    return it == tok_to_id.end() ? -1 : it->second;
}

int32_t Tokenizer::language_token(const std::string& code) const { return token_to_id("<|" + code + "|>"); }

// ------------------------------------------------------------------------------------------ encode
// BPE of one pre-tokenized piece (`tokenizers` models/bpe/word.rs merge_all): start from one symbol per byte-level
// character, repeatedly apply the applicable merge of lowest rank (leftmost first).
void Tokenizer::bpe_word(const std::string& piece, std::vector<int32_t>& out) const {
    auto ic = cache.find(piece);
    if (ic != cache.end()) { out.insert(out.end(), ic->second.begin(), ic->second.end()); return; }
    const ByteMap& bm = byte_map();
    std::vector<int32_t> sym;
    sym.reserve(piece.size());
    for (unsigned char b : piece) {
        auto it = tok_to_id.find(bm.byte_to_str[b]);
        WRS_REQUIRE(it != tok_to_id.end(), "tokenizer: byte 0x%02x has no vocabulary entry", (unsigned)b);
        sym.push_back(it->second);
    }
    while (sym.size() > 1) {
        int32_t best_rank = INT32_MAX, best_pos = -1, best_new = -1;
        for (size_t i = 0; i + 1 < sym.size(); ++i) {
            auto it = merges.find(((uint64_t)(uint32_t)sym[i] << 32) | (uint32_t)sym[i + 1]);
            if (it != merges.end() && it->second.first < best_rank) { best_rank = it->second.first; best_pos = (int32_t)i; best_new = it->second.second; }
        }
        if (best_pos < 0) break;
        sym[best_pos] = best_new;
        sym.erase(sym.begin() + best_pos + 1);
    }
    if (cache.size() < 100000) cache.emplace(piece, sym);
    out.insert(out.end(), sym.begin(), sym.end());
}

// ByteLevel pre-tokenizer pattern (`tokenizers` pre_tokenizers/byte_level.rs):
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
void Tokenizer::encode_plain(const std::string& text, std::vector<int32_t>& out) const {
    if (text.empty()) return;
    std::vector<uint32_t> cp, off;
    decode_utf8_lossy(text, cp, &off);
    const size_t n = cp.size();
    auto other = [&](uint32_t c) { return !is_space(c) && !is_letter(c) && !is_number(c); };
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        if (cp[i] == '\'' && i + 1 < n) {
            const uint32_t a = cp[i + 1], b = i + 2 < n ? cp[i + 2] : 0;
            if (a == 's' || a == 't' || a == 'm' || a == 'd') j = i + 2;
            else if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) j = i + 3;
        }
        if (j == i) {
            const size_t s = (cp[i] == ' ' && i + 1 < n) ? i + 1 : i;  // after the optional single space
            if (is_letter(cp[s])) { j = s; while (j < n && is_letter(cp[j])) ++j; }
            else if (is_number(cp[s])) { j = s; while (j < n && is_number(cp[j])) ++j; }
            else if (other(cp[s])) { j = s; while (j < n && other(cp[j])) ++j; }
            else {  // whitespace run: \s+(?!\S) keeps the last blank for the next piece unless the run ends the text
                j = i;
                while (j < n && is_space(cp[j])) ++j;
                if (j < n && j - i > 1) --j;
            }
        }
        std::string piece = text.substr(off[i], off[j] - off[i]);
        // a lossy-decoded ill-formed byte is encoded as U+FFFD, as Rust would have handed it over
        bool lossy = false;
        for (size_t k = i; k < j; ++k)
            if (cp[k] == 0xFFFD) { lossy = true; break; }
        if (lossy) {
            piece.clear();
            for (size_t k = i; k < j; ++k) append_utf8(piece, cp[k]);
        }
        bpe_word(piece, out);
        i = j;
    }
}

// `TokenizerImpl::encode`: added tokens are cut out of the text first (leftmost-longest), the rest goes through the
// pre-tokenizer and the BPE model.  The ByteLevel post-processor only touches offsets.
std::vector<int32_t> Tokenizer::encode(const std::string& text) const {
    std::vector<int32_t> out;
    size_t pos = 0, start = 0;
    const size_t n = text.size();
    while (pos < n) {
        const Added* hit = nullptr;
        if (text[pos] == '<' || !added.empty()) {
            for (const Added& a : added) {
                if (a.content.empty() || a.content[0] != text[pos] || a.content.size() > n - pos) continue;
                if (text.compare(pos, a.content.size(), a.content) == 0 && (!hit || a.content.size() > hit->content.size())) hit = &a;
            }
        }
        if (hit) {
            encode_plain(text.substr(start, pos - start), out);
            out.push_back(hit->id);
            pos += hit->content.size();
            start = pos;
        } else ++pos;
    }
    encode_plain(text.substr(start), out);
    return out;
}

// ------------------------------------------------------------------------------------------ decode
std::string Tokenizer::decode(const int64_t* tokens, int64_t n) const {
    const ByteMap& bm = byte_map();
    std::string bytes;
    std::vector<uint32_t> cps;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t id = tokens[i];
        if (id < 0 || (size_t)id >= id_to_tok.size()) continue;  // timestamp ids: id_to_token() is None -> dropped
        const std::string& t = id_to_tok[(size_t)id];
        if (t.empty()) continue;
        cps.clear();
        decode_utf8_lossy(t, cps, nullptr);
        std::string mapped;
        bool all = true;
        for (uint32_t c : cps) {
            auto it = bm.cp_to_byte.find(c);
            if (it == bm.cp_to_byte.end()) { all = false; break; }
            mapped.push_back((char)it->second);
        }
        bytes += all ? mapped : t;
    }
    cps.clear();
    decode_utf8_lossy(bytes, cps, nullptr);
    std::string out;
    out.reserve(bytes.size());
    for (uint32_t c : cps) append_utf8(out, c);
    return out;
}

// ------------------------------------------------------------------------------------------ whisper helpers
std::vector<int32_t> Tokenizer::non_speech_tokens() const {
    static const char* const symbols[] = {"\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
                                          "「", "」", "『", "』", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))",
                                          "(((", ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪"};
    static const char* const misc[] = {"♩", "♪", "♫", "♬", "♭", "♮", "♯"};
    std::vector<int32_t> r;
    r.push_back(encode(" -")[0]);
    r.push_back(encode(" '")[0]);
    for (const char* s : symbols) {
        std::vector<int32_t> t = encode(s);
        if (t.size() == 1) r.push_back(t[0]);
        t = encode(std::string(" ") + s);
        if (t.size() == 1) r.push_back(t[0]);
    }
    for (const char* s : misc) {
        r.push_back(encode(s)[0]);
        r.push_back(encode(std::string(" ") + s)[0]);
    }
    std::sort(r.begin(), r.end());
    return r;
}

std::vector<int32_t> Tokenizer::sequence_sot(int32_t language_id, int32_t task_id) const {
    std::vector<int32_t> r{ids.sot};
    if (language_id >= 0) r.push_back(language_id);
    if (task_id >= 0) r.push_back(task_id);
    return r;
}

}  // namespace wrs
