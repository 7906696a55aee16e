// Host-side WordPiece tokeniser feeding the encoder ("at GPU rates the tokeniser is the end-to-end limiter").
//
// What sentence-transformers runs before `encode` for the models of this path is the HF `tokenizers` pipeline
//   BertNormalizer(clean_text, handle_chinese_chars, lowercase[/strip_accents]) -> BertPreTokenizer (whitespace + punctuation split)
//   -> WordPiece("##", max_input_chars_per_word) -> "<bos> $A <eos>" -> truncation to max_seq_length
// (transformers models/mpnet/tokenization_mpnet.py:108-163, models/bert/tokenization_bert.py).  On pure-ASCII text every stage of
// that pipeline is table-free: no Unicode categories, no NFD, no CJK.  This file implements exactly that ASCII restriction,
// multi-threaded, writing a padded int32 id matrix + lengths directly (no per-token Python objects).  Non-ASCII text is handled per
// whitespace-delimited SEGMENT through a cache of the reference pipeline's own output for that segment (see WordPiece::seg_cache);
// texts it must not touch at all (an occurrence of one of the tokenizer's added-token strings, a non-ASCII run longer than 512
// bytes) are FLAGGED so the caller routes them through the reference pipeline.  Results are identical to the HF pipeline on
// every input (tests/test_host_cli.py fuzzes both, ASCII and Unicode).
//
// Plain C ABI, no device code: built with g++ into libarx_host.so.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Table {                       // open-addressing map: byte string -> id
    std::vector<int32_t> slot_id;    // -1 = empty
    std::vector<uint32_t> slot_off, slot_len;
    std::string pool;
    uint32_t mask = 0;
    int max_len = 0;
    static uint64_t hash(const char* p, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        return h ^ (h >> 29);
    }
    void build(const std::vector<std::pair<std::string, int32_t>>& items) {
        size_t cap = 16;
        while (cap < items.size() * 2 + 2) cap <<= 1;
        mask = (uint32_t)cap - 1;
        slot_id.assign(cap, -1); slot_off.assign(cap, 0); slot_len.assign(cap, 0);
        for (auto& it : items) {
            uint32_t s = (uint32_t)hash(it.first.data(), it.first.size()) & mask;
            bool dup = false;
            while (slot_id[s] >= 0) {
                if (slot_len[s] == it.first.size() && !memcmp(pool.data() + slot_off[s], it.first.data(), it.first.size())) { dup = true; break; }
                s = (s + 1) & mask;
            }
            if (dup) continue;                                   // first occurrence wins, like a dict built with setdefault
            slot_id[s] = it.second; slot_off[s] = (uint32_t)pool.size(); slot_len[s] = (uint32_t)it.first.size();
            pool += it.first;
            max_len = std::max(max_len, (int)it.first.size());
        }
    }
    int32_t find(const char* p, size_t n) const {
        if ((int)n > max_len) return -1;
        uint32_t s = (uint32_t)hash(p, n) & mask;
        while (slot_id[s] >= 0) {
            if (slot_len[s] == n && !memcmp(pool.data() + slot_off[s], p, n)) return slot_id[s];
            s = (s + 1) & mask;
        }
        return -1;
    }
};

struct WordPiece {
    Table head, cont;                // whole-word-start pieces; continuation pieces keyed WITHOUT their "##"
    int32_t unk, bos, eos, pad, lowercase, max_chars;
    std::vector<std::string> triggers;
    // Non-ASCII SEGMENTS (maximal runs between ASCII whitespace that contain a byte >= 0x80): every stage of the pipeline is local
    // to such a segment (per-character normalisation, NFD inside combining sequences, whitespace/punctuation split, per-word
    // WordPiece), so tokens(text) = concatenation of tokens(segment).  The reference pipeline tokenises each distinct segment ONCE
    // (the caller resolves misses and feeds them back through arx_wp_cache_add); afterwards "α", "–", "naïve", ... cost a hash lookup.
    std::unordered_map<std::string, std::vector<int32_t>> seg_cache;
    std::vector<std::string> misses;             // distinct unresolved segments of the last arx_wp_encode call
    std::mutex miss_mu;
};
constexpr int64_t SEG_MAX_BYTES = 512;           // longer non-ASCII runs (unsegmented CJK prose, ...) send the whole text to the fallback

inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_removed(unsigned char c) { return c == 0 || (c < 0x20 && c != '\t' && c != '\n' && c != '\r') || c == 0x7f; }   // clean_text: NUL + Cc
inline bool is_punct(unsigned char c) { return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126); }

// one word (no whitespace, no punctuation, already lower-cased, control bytes removed) -> pieces appended to out (at most room)
inline void wordpiece(const WordPiece& w, const char* p, int n, int32_t* out, int& cnt, int room) {
    if (cnt >= room) return;
    if (n > w.max_chars) { out[cnt++] = w.unk; return; }
    int32_t tmp[128];
    int nt = 0, start = 0;
    while (start < n) {
        const Table& t = start ? w.cont : w.head;
        int end = std::min(n, start + t.max_len), id = -1;
        for (; end > start; --end) { id = t.find(p + start, end - start); if (id >= 0) break; }
        if (id < 0) { out[cnt++] = w.unk; return; }             // any unmatched position: the whole word is unknown
        tmp[nt++] = id; start = end;
    }
    for (int i = 0; i < nt && cnt < room; ++i) out[cnt++] = tmp[i];
}

// fallback codes: 0 = tokenised here, 1 = whole text must take the reference pipeline, 2 = has segments missing from the cache
void encode_one(const WordPiece& w, const char* s, int64_t n, int max_len, int32_t* row, int32_t* len_out, uint8_t* fallback,
                std::vector<std::string>& local_misses) {
    for (auto& t : w.triggers)
        if (!t.empty() && n >= (int64_t)t.size() && std::search(s, s + n, t.begin(), t.end()) != s + n) { *fallback = 1; *len_out = 0; return; }
    const int room = std::max(0, max_len - 2);                   // truncation keeps the first max_len-2 pieces
    int cnt = 0;
    int32_t* out = row + 1;
    char word[104];
    int wl = 0;                                                  // chars of the current word; > 100 only needs counting
    bool over = false, missing = false;
    auto flush = [&]() {
        if (wl > 0 || over) {
            if (over) { if (cnt < room) out[cnt++] = w.unk; }
            else wordpiece(w, word, wl, out, cnt, room);
        }
        wl = 0; over = false;
    };
    int64_t i = 0;
    while (i < n && (cnt < room || missing)) {
        if (is_space((unsigned char)s[i])) { ++i; continue; }
        int64_t e = i;
        bool ascii = true;
        while (e < n && !is_space((unsigned char)s[e])) { ascii &= ((unsigned char)s[e] < 0x80); ++e; }
        if (ascii) {
            if (!missing)
                for (int64_t k = i; k < e && cnt < room; ++k) {
                    unsigned char c = (unsigned char)s[k];
                    if (is_removed(c)) continue;
                    if (is_punct(c)) {
                        flush();
                        if (cnt < room) { const char pc = (char)c; wordpiece(w, &pc, 1, out, cnt, room); }
                        continue;
                    }
                    if (w.lowercase && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
                    if (wl < w.max_chars && wl < 100) word[wl++] = (char)c; else over = true;
                }
            if (cnt < room) flush(); else { wl = 0; over = false; }
        } else {
            if (e - i > SEG_MAX_BYTES) { *fallback = 1; *len_out = 0; return; }
            const std::string key(s + i, (size_t)(e - i));
            auto it = w.seg_cache.find(key);
            if (it == w.seg_cache.end()) { missing = true; local_misses.push_back(key); }
            else if (!missing)
                for (size_t k = 0; k < it->second.size() && cnt < room; ++k) out[cnt++] = it->second[k];
        }
        i = e;
    }
    if (missing) { *fallback = 2; *len_out = 0; return; }
    *fallback = 0;
    row[0] = w.bos;
    row[1 + cnt] = w.eos;
    for (int k = cnt + 2; k < max_len; ++k) row[k] = w.pad;
    *len_out = cnt + 2;
}

}  // namespace

extern "C" {

// vocab: n_vocab strings (blob + offsets[n_vocab+1]), id = index.  triggers: strings whose presence in a text sends it to the fallback.
int32_t arx_wp_create(const char* blob, const int64_t* off, int32_t n_vocab, int32_t unk, int32_t bos, int32_t eos, int32_t pad,
                      int32_t lowercase, int32_t max_chars, const char* trig_blob, const int64_t* trig_off, int32_t n_trig, void** out) {
    if (!blob || !off || !out || n_vocab <= 0 || max_chars <= 0 || max_chars > 100) return -1;
    auto* w = new WordPiece();
    std::vector<std::pair<std::string, int32_t>> head, cont;
    for (int32_t i = 0; i < n_vocab; ++i) {
        std::string t(blob + off[i], (size_t)(off[i + 1] - off[i]));
        if (t.empty()) continue;
        if (t.size() > 2 && t[0] == '#' && t[1] == '#') cont.emplace_back(t.substr(2), i);
        else head.emplace_back(t, i);
    }
    w->head.build(head); w->cont.build(cont);
    w->unk = unk; w->bos = bos; w->eos = eos; w->pad = pad; w->lowercase = lowercase; w->max_chars = max_chars;
    for (int32_t i = 0; i < n_trig; ++i) w->triggers.emplace_back(trig_blob + trig_off[i], (size_t)(trig_off[i + 1] - trig_off[i]));
    *out = w;
    return 0;
}

void arx_wp_destroy(void* h) { delete static_cast<WordPiece*>(h); }

// texts: n strings (blob + offsets[n+1]).  ids: [n, max_len] int32 (padded with pad id), lens: [n], fallback: [n] (1 = not tokenised
// here: non-ASCII or contains an added-token string; its row is untouched and its length 0).
int32_t arx_wp_encode(void* h, const char* blob, const int64_t* off, int64_t n, int32_t max_len, int32_t* ids, int32_t* lens,
                      uint8_t* fallback, int32_t n_threads) {
    if (!h || !off || !ids || !lens || !fallback || n < 0 || max_len < 2) return -1;
    WordPiece& w = *static_cast<WordPiece*>(h);
    w.misses.clear();
    if (n == 0) return 0;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 1, (n + 255) / 256));
    std::atomic<int64_t> next(0);
    std::unordered_set<std::string> seen;
    auto work = [&]() {
        std::vector<std::string> local;
        for (;;) {
            const int64_t b = next.fetch_add(256);
            if (b >= n) break;
            const int64_t e = std::min(n, b + 256);
            for (int64_t i = b; i < e; ++i)
                encode_one(w, blob + off[i], off[i + 1] - off[i], max_len, ids + i * (int64_t)max_len, lens + i, fallback + i, local);
        }
        if (!local.empty()) {
            std::lock_guard<std::mutex> g(w.miss_mu);
            for (auto& m : local) if (seen.insert(m).second) w.misses.push_back(std::move(m));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return 0;
}

// Distinct non-ASCII segments the last arx_wp_encode call could not resolve (fallback code 2): sizes, then the strings themselves.
int32_t arx_wp_miss_count(void* h, int64_t* n_strings, int64_t* n_bytes) {
    if (!h || !n_strings || !n_bytes) return -1;
    const WordPiece& w = *static_cast<WordPiece*>(h);
    int64_t b = 0;
    for (auto& m : w.misses) b += (int64_t)m.size();
    *n_strings = (int64_t)w.misses.size(); *n_bytes = b;
    return 0;
}
int32_t arx_wp_miss_fetch(void* h, char* blob, int64_t* off /* [n_strings+1] */) {
    if (!h || !off) return -1;
    const WordPiece& w = *static_cast<WordPiece*>(h);
    int64_t b = 0, i = 0;
    for (auto& m : w.misses) { off[i++] = b; if (!m.empty()) memcpy(blob + b, m.data(), m.size()); b += (int64_t)m.size(); }
    off[i] = b;
    return 0;
}
// Teach the tokenizer the reference pipeline's pieces for segments (no specials, no truncation).  Not concurrent with arx_wp_encode.
int32_t arx_wp_cache_add(void* h, const char* seg_blob, const int64_t* seg_off, int64_t n, const int32_t* ids, const int64_t* ids_off) {
    if (!h || !seg_off || !ids_off || n < 0) return -1;
    WordPiece& w = *static_cast<WordPiece*>(h);
    for (int64_t i = 0; i < n; ++i)
        w.seg_cache[std::string(seg_blob + seg_off[i], (size_t)(seg_off[i + 1] - seg_off[i]))] =
            std::vector<int32_t>(ids + ids_off[i], ids + ids_off[i + 1]);
    return 0;
}
int64_t arx_wp_cache_size(void* h) { return h ? (int64_t)static_cast<WordPiece*>(h)->seg_cache.size() : -1; }

int32_t arx_wp_version() { return 2; }

}  // extern "C"
