// falcon_tokenizer.hip -- host-only: the Falcon BPE tokenizer, same results as the reference's
// falcon_tokenize (libfalcon.cpp:2594-3035 with cmpnct_unicode.cpp) on the vocabulary + merges of a GGCC v10 file.
//
// Three steps, as in the reference:
//   1. pre-split (bpe_gpt2_preprocess, libfalcon.cpp:2797-2993): special tokens are cut out verbatim; the rest is split
//      GPT-2 style by a hand-written scanner over UTF-8 characters classified as letter / digit / whitespace / other.
//      The scanner is restated here with its observable quirks, because they decide token ids:
//        - a "'" followed by s/t/m/d splits off two characters; a "'" followed by r, v or l -- OR by anything whose
//          second successor is e or l -- splits off three (the reference tests `next == 'r' || next_next == 'e'`);
//        - the very last character of the text is appended to the word in progress even when it would start a new one;
//        - a word that starts with a space continues as a "special" run only if the next character is not a letter,
//          digit or whitespace, and so on: see scan_words.
//   2. every word is mapped byte -> printable code point (GPT-2's bytes_to_unicode) and merged by rank: the lowest rank
//      first, ties to the leftmost pair (libfalcon.cpp:2627-2716); special-token words are not merged;
//   3. every merged piece is mapped back to bytes and looked up in the vocabulary; a piece that is not a token falls
//      back to one token per byte (libfalcon.cpp:2735-2752).
// No GPU work: the file is part of libggml_hip.so so that a drop-in user finds falcon_tokenize's replacement next to eval.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>
#include "fq_unicode_tables.h"
#include "../../include/falcon-hip.h"

namespace {

enum cp_class { CP_LETTER, CP_DIGIT, CP_SPACE, CP_OTHER };

template <size_t N> bool in_runs(int c, const int (&first)[N], const int (&extra)[N]) {
    const int * it = std::upper_bound(first, first + N, c);  // the last run that starts at or before c
    if (it == first) return false;
    const size_t i = (size_t)(it - first) - 1;
    return c - first[i] <= extra[i];
}
cp_class classify(int c) {                                   // cmpnct_unicode.cpp:98-115 (letters are tested first)
    if (c < 0) return CP_OTHER;
    if (in_runs(c, fq_cp_letters_first, fq_cp_letters_extra)) return CP_LETTER;
    if (in_runs(c, fq_cp_digits_first, fq_cp_digits_extra)) return CP_DIGIT;
    if (in_runs(c, fq_cp_spaces_first, fq_cp_spaces_extra)) return CP_SPACE;
    return CP_OTHER;
}
// lead byte -> sequence length, as the reference counts it (cmpnct_unicode.cpp:135-147): continuation bytes and 0xF8..0xFF
// are not rejected
int seq_len(unsigned char c) {
    if ((c & 0x80) == 0) return 1;
    if ((c & 0xE0) == 0xC0) return 2;
    if ((c & 0xF0) == 0xE0) return 3;
    if ((c & 0xF0) == 0xF0) return 4;
    return 1;
}
// code point of one sequence; a lone byte >= 0x80 is a negative `char` in the reference and belongs to no class
int code_point(const std::string & s) {
    const size_t n = s.size();
    const auto b = [&](size_t i) { return (int)(signed char) s[i]; };
    if (n == 1) return b(0);
    if (n == 2) return ((b(0) & 0x1F) << 6) | (b(1) & 0x3F);
    if (n == 3) return ((b(0) & 0x0F) << 12) | ((b(1) & 0x3F) << 6) | (b(2) & 0x3F);
    if (n == 4) return ((b(0) & 0x07) << 18) | ((b(1) & 0x3F) << 12) | ((b(2) & 0x3F) << 6) | (b(3) & 0x3F);
    return 0;
}

struct uchar { std::string s; cp_class cls = CP_OTHER; size_t offset = 0; };

std::vector<std::string> split_chars(const std::string & text) {
    std::vector<std::string> out;
    for (size_t i = 0; i < text.size();) {
        const size_t n = std::min(text.size() - i, (size_t) seq_len((unsigned char) text[i]));
        out.emplace_back(text, i, n);
        i += n;
    }
    return out;
}

// GPT-2's byte <-> printable code point map: the bytes that print ('!'..'~', 0xA1..0xAC, 0xAE..0xFF) stand for themselves,
// the others are numbered 0x100, 0x101, ... in byte order
struct byte_map {
    std::string enc[256];
    std::unordered_map<std::string, unsigned char> dec;
    byte_map() {
        int next = 0x100;
        for (int b = 0; b < 256; ++b) {
            const bool keeps = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
            const int cp = keeps ? b : next++;
            std::string u;
            if (cp < 0x80) u.push_back((char) cp);
            else { u.push_back((char)(0xC0 | (cp >> 6))); u.push_back((char)(0x80 | (cp & 0x3F))); }
            enc[b] = u; dec[u] = (unsigned char) b;
        }
    }
};
const byte_map & bytes() { static const byte_map m; return m; }

}  // namespace

struct falcon_hip_vocab {
    std::vector<std::string> id_to_token;
    std::unordered_map<std::string, int32_t> token_to_id;
    std::map<std::pair<std::string, std::string>, int> rank;
    std::map<std::string, int32_t> special;                  // ordered: the reference walks a std::map (first match in key order)
    std::string error;
};

namespace {

// ---- step 1
std::vector<std::string> scan_words(const falcon_hip_vocab & v, const std::string & text) {
    std::vector<uchar> cs;
    {
        size_t off = 0;
        for (const std::string & s : split_chars(text)) { uchar u; u.s = s; u.cls = classify(code_point(s)); u.offset = off; off += s.size(); cs.push_back(u); }
    }
    size_t shortest_special = 0;
    for (const auto & kv : v.special) shortest_special = shortest_special ? std::min(shortest_special, kv.first.size()) : kv.first.size();
    const uchar none;                                        // what the reference sees beyond the end: empty, no class
    enum { RUN_NONE, RUN_LETTERS, RUN_DIGITS, RUN_OTHER, RUN_SPACES } run = RUN_NONE;
    std::vector<std::string> words;
    std::string word;
    const size_t text_len = strlen(text.c_str());            // (the reference measures what is left with strlen)
    for (size_t i = 0; i < cs.size(); ++i) {
        const uchar & c = cs[i];
        const uchar & n1 = i + 1 < cs.size() ? cs[i + 1] : none;
        const uchar & n2 = i + 2 < cs.size() ? cs[i + 2] : none;
        const size_t remain = c.offset < text_len ? text_len - c.offset : 0;
        // special tokens, verbatim
        bool took_special = false;
        if (remain >= shortest_special) {
            for (const auto & kv : v.special) {
                const std::string & sp = kv.first;
                if (remain < sp.size() || text.compare(c.offset, sp.size(), sp) != 0) continue;
                if (!word.empty()) { words.push_back(word); word.clear(); run = RUN_NONE; }
                words.push_back(sp);
                size_t left = sp.size();                     // skip the characters the token covers
                while (left && i < cs.size()) { left -= std::min(left, cs[i].s.size()); ++i; }
                --i;
                took_special = true;
                break;
            }
        }
        if (took_special) continue;
        // contractions (the run state is NOT reset, as in the reference)
        if (remain >= 2 && c.s == "'" && (n1.s == "s" || n1.s == "t" || n1.s == "m" || n1.s == "d")) {
            if (!word.empty()) words.push_back(word);
            words.push_back(c.s + n1.s);
            word.clear();
            i += 1;
            continue;
        }
        if (remain >= 3 && c.s == "'" && (n1.s == "r" || n1.s == "v" || n1.s == "l" || n2.s == "e" || n2.s == "l")) {
            if (!word.empty()) words.push_back(word);
            words.push_back(c.s + n1.s + n2.s);
            word.clear();
            i += 2;
            continue;
        }
        bool split = false;
        const bool lead_space = word.empty() && c.s == " ";
        if (run == RUN_NONE) {
            if (c.cls == CP_LETTER || (lead_space && n1.cls == CP_LETTER)) run = RUN_LETTERS;
            else if (c.cls == CP_DIGIT || (lead_space && n1.cls == CP_DIGIT)) run = RUN_DIGITS;
            else if ((c.cls != CP_LETTER && c.cls != CP_DIGIT && c.cls != CP_SPACE) ||
                     (lead_space && n1.cls != CP_LETTER && n1.cls != CP_DIGIT && n1.cls != CP_SPACE)) run = RUN_OTHER;
            else if (c.cls == CP_SPACE && n1.cls == CP_SPACE) run = RUN_SPACES;
            else if (c.cls == CP_SPACE) split = true;
        } else {
            if (run == RUN_LETTERS && c.cls != CP_LETTER) split = true;
            else if (run == RUN_DIGITS && c.cls != CP_DIGIT) split = true;
            else if (run == RUN_OTHER && (c.cls == CP_LETTER || c.cls == CP_DIGIT || c.cls == CP_SPACE)) split = true;
            else if (run == RUN_SPACES && n1.cls != CP_SPACE) split = true;
        }
        if (n1.s.empty()) { split = true; word += c.s; }     // the last character joins the word in progress
        if (split) {
            if (!word.empty()) words.push_back(word);
            word = c.s;
            run = RUN_NONE;
        } else {
            word += c.s;
        }
    }
    return words;
}

// ---- step 2: merge one word (already in the printable alphabet) by rank
struct piece { int prev, next; size_t pos, len; };
struct bigram { int left, right, rank; std::string text; };
struct bigram_later { bool operator()(const bigram & a, const bigram & b) const { return a.rank > b.rank || (a.rank == b.rank && a.left > b.left); } };

void merge_word(const falcon_hip_vocab & v, const std::string & w, bool is_special, std::vector<std::string> & out) {
    std::vector<piece> p;
    if (is_special) {
        p.push_back({ -1, -1, 0, w.size() });
    } else {
        for (size_t off = 0; off < w.size();) {
            const size_t n = std::min(w.size() - off, (size_t) seq_len((unsigned char) w[off]));
            const int idx = (int) p.size();
            p.push_back({ idx - 1, off + n == w.size() ? -1 : idx + 1, off, n });
            off += n;
        }
    }
    std::priority_queue<bigram, std::vector<bigram>, bigram_later> q;
    const auto text_of = [&](int i) { return w.substr(p[(size_t) i].pos, p[(size_t) i].len); };
    const auto offer = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        const std::string a = text_of(l), b = text_of(r);
        const auto it = v.rank.find(std::make_pair(a, b));
        if (it == v.rank.end()) return;
        q.push(bigram{ l, r, it->second, a + b });
    };
    if (!is_special) for (size_t i = 1; i < p.size(); ++i) offer((int) i - 1, (int) i);
    while (!q.empty()) {
        const bigram b = q.top(); q.pop();
        piece & L = p[(size_t) b.left];
        piece & R = p[(size_t) b.right];
        if (L.len == 0 || R.len == 0 || text_of(b.left) + text_of(b.right) != b.text) continue;      // stale
        L.len += R.len; R.len = 0;
        L.next = R.next;
        if (R.next >= 0) p[(size_t) R.next].prev = b.left;
        offer(L.prev, b.left);
        offer(b.left, L.next);
    }
    for (const piece & x : p) if (x.len) out.push_back(w.substr(x.pos, x.len));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- C ABI
extern "C" falcon_hip_vocab * falcon_hip_vocab_load_ggcc(const char * path) {
    falcon_hip_vocab * v = new falcon_hip_vocab();
    FILE * f = fopen(path, "rb");
    if (!f) { v->error = "cannot open file"; return v; }
    const auto rd_u32 = [&](uint32_t & x) { return fread(&x, 4, 1, f) == 1; };
    const auto rd_str = [&](std::string & s, uint32_t n) { s.resize(n); return n == 0 || fread(&s[0], 1, n, f) == n; };
    uint32_t magic = 0, version = 0, h[8] = {0};
    bool ok = rd_u32(magic) && rd_u32(version);
    for (int i = 0; ok && i < 8; ++i) ok = rd_u32(h[i]);
    if (!ok || magic != 0x67676363u || version != 10) { v->error = "not a GGCC v10 file"; fclose(f); return v; }
    const uint32_t n_vocab = h[0];
    v->id_to_token.resize(n_vocab);
    for (uint32_t i = 0; ok && i < n_vocab; ++i) {
        uint32_t len = 0; float score = 0.0f;
        ok = rd_u32(len) && len < (1u << 20) && rd_str(v->id_to_token[i], len) && fread(&score, 4, 1, f) == 1;
        if (ok) v->token_to_id[v->id_to_token[i]] = (int32_t) i;
    }
    uint32_t n_merges = 0;
    ok = ok && rd_u32(n_merges);
    for (uint32_t i = 0; ok && i < n_merges; ++i) {
        uint32_t l1 = 0, l2 = 0; std::string a, b;
        ok = rd_u32(l1) && l1 < (1u << 20) && rd_str(a, l1) && rd_u32(l2) && l2 < (1u << 20) && rd_str(b, l2);
        if (ok) v->rank.emplace(std::make_pair(a, b), (int) i);         // (emplace: the first of two equal pairs keeps its rank)
    }
    fclose(f);
    if (!ok) { v->error = "truncated vocabulary"; return v; }
    if (n_vocab < 12) { v->error = "vocabulary too small (ids 0..11 are the special tokens)"; return v; }
    if (n_vocab == 65025 && v->id_to_token[65024] == "[PAD]") {          // libfalcon.cpp:862-868: the padding token is dropped
        v->token_to_id.erase("[PAD]");
        v->id_to_token.resize(65024);
    }
    // special tokens: ids 0..11 and 65024.. (libfalcon.cpp:320-326)
    for (size_t i = 0; i < 12; ++i) v->special[v->id_to_token[i]] = (int32_t) i;
    for (size_t i = 65024; i < v->id_to_token.size(); ++i) v->special[v->id_to_token[i]] = (int32_t) i;
    return v;
}
extern "C" const char * falcon_hip_vocab_error(const falcon_hip_vocab * v) { return v->error.empty() ? nullptr : v->error.c_str(); }
extern "C" void falcon_hip_vocab_free(falcon_hip_vocab * v) { delete v; }
extern "C" int falcon_hip_vocab_size(const falcon_hip_vocab * v) { return (int) v->id_to_token.size(); }
extern "C" int falcon_hip_vocab_merges(const falcon_hip_vocab * v) { return (int) v->rank.size(); }
extern "C" int falcon_hip_token_to_bytes(const falcon_hip_vocab * v, int32_t id, const char ** bytes) {
    if (id < 0 || (size_t) id >= v->id_to_token.size()) return -1;
    *bytes = v->id_to_token[(size_t) id].data();
    return (int) v->id_to_token[(size_t) id].size();
}
extern "C" int32_t falcon_hip_token_bos(void) { return 11; }     // libfalcon.cpp:4684-4690
extern "C" int32_t falcon_hip_token_eos(void) { return 11; }

// falcon_tokenize (libfalcon.cpp:4623-4641): the ids of `text` (bos first when add_bos and the text is not empty); returns
// their number, or minus that number when n_max is too small (nothing written then)
extern "C" int falcon_hip_tokenize(const falcon_hip_vocab * v, const char * text_c, int32_t * tokens, int n_max, int add_bos) {
    std::vector<int32_t> ids;
    const std::string text(text_c);
    if (!text.empty()) {
        if (add_bos) ids.push_back(falcon_hip_token_bos());
        std::vector<std::string> pieces;
        for (const std::string & word : scan_words(*v, text)) {
            const bool is_special = v->special.count(word) != 0;
            std::string enc;
            for (unsigned char b : word) enc += bytes().enc[b];
            // (a special token's word is merged as itself: the reference compares the ENCODED word with the token, so a
            //  token containing a byte outside '!'..'~' is not recognised at this point)
            const bool verbatim = v->special.count(enc) != 0;
            (void) is_special;
            merge_word(*v, enc, verbatim, pieces);
        }
        for (const std::string & pc : pieces) {
            std::string raw;
            for (const std::string & ch : split_chars(pc)) {
                const auto it = bytes().dec.find(ch);
                raw.push_back(it == bytes().dec.end() ? '\0' : (char) it->second);
            }
            const auto tok = v->token_to_id.find(raw);
            if (tok != v->token_to_id.end()) { ids.push_back(tok->second); continue; }
            for (char b : raw) {                             // byte fallback
                const auto bt = v->token_to_id.find(std::string(1, b));
                if (bt == v->token_to_id.end()) { fprintf(stderr, "falcon-hip: tokenizer: byte 0x%02x is not in the vocabulary\n", (unsigned)(unsigned char) b); return INT32_MIN; }
                ids.push_back(bt->second);
            }
        }
    }
    if ((int) ids.size() > n_max) return -(int) ids.size();
    for (size_t i = 0; i < ids.size(); ++i) tokens[i] = ids[i];
    return (int) ids.size();
}
