// tokenizer.cpp — BERT (uncased) WordPiece tokenizer behind the C ABI.
//
// In the reference the text -> token step happens inside Ollama (llama.cpp's WPM tokenizer)
// for every chunk chunk_text() produces (app/main.py:2160-2170 -> 225-237).  Here it is host
// C++ so that ingest is not bottlenecked on Python: BasicTokenizer (clean, CJK spacing,
// lower-case, NFD + strip Mn, punctuation split) + greedy longest-match-first WordPiece,
// [CLS] ... [SEP], truncated to max_len.  Code-point classes and case/accent folds come from
// tables generated out of Python's unicodedata (tools/gen_unicode_tables.py), so the output
// matches rassengine_amd.encoder.WordPieceTokenizer / transformers.BertTokenizer
// (tests/test_tokenizer_cpp.py).  Batch encoding fans out over std::thread.

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/rass_engine.h"

extern "C" void rassint_set_last_error(const char* msg);

namespace {

#include "unicode_tables.inc"

constexpr uint8_t kPunct = 1, kSpace = 2, kControl = 4, kCased = 8, kIgnorable = 16;

int tfail(int code, const std::string& msg) {
    rassint_set_last_error(msg.c_str());
    return code;
}

uint8_t cp_flags(uint32_t cp) {
    if (cp < 0x80) {
        uint8_t f = 0;
        for (uint8_t c : kAsciiCased)
            if (c == cp) f |= kCased;
        for (uint8_t c : kAsciiIgnorable)
            if (c == cp) f |= kIgnorable;
        if ((cp >= 33 && cp <= 47) || (cp >= 58 && cp <= 64) || (cp >= 91 && cp <= 96) || (cp >= 123 && cp <= 126))
            return f | kPunct;
        if (cp == ' ') return kSpace;
        if (cp == '\t' || cp == '\n' || cp == '\r') return kSpace;  // BERT treats these as whitespace
        if (cp < 32 || cp == 127) return kControl;
        return f;
    }
    const size_t n = sizeof(kCpFlags) / sizeof(kCpFlags[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kCpFlags[mid].cp < cp) lo = mid + 1; else hi = mid;
    }
    return (lo < n && kCpFlags[lo].cp == cp) ? kCpFlags[lo].flags : 0;
}

// lower-case + NFD + drop Mn of one code point -> 0..3 code points
int cp_fold(uint32_t cp, uint32_t out[3]) {
    if (cp < 0x80) {
        out[0] = (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
        return 1;
    }
    const size_t n = sizeof(kCpFold) / sizeof(kCpFold[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kCpFold[mid].cp < cp) lo = mid + 1; else hi = mid;
    }
    if (lo < n && kCpFold[lo].cp == cp) {
        for (int i = 0; i < kCpFold[lo].n; ++i) out[i] = kCpFold[lo].to[i];
        return kCpFold[lo].n;
    }
    out[0] = cp;
    return 1;
}

bool is_cjk(uint32_t cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
           (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
           (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}

void utf8_append(std::string& s, uint32_t cp) {
    if (cp < 0x80) {
        s.push_back((char)cp);
    } else if (cp < 0x800) {
        s.push_back((char)(0xC0 | (cp >> 6)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18)));
        s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
        s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// decode UTF-8; malformed bytes become U+FFFD (which the cleaner drops, like Python's errors="replace" + clean)
void utf8_decode(const char* p, int64_t len, std::vector<uint32_t>& out) {
    out.clear();
    out.reserve((size_t)len);
    for (int64_t i = 0; i < len;) {
        const unsigned char c = (unsigned char)p[i];
        uint32_t cp = 0xFFFD;
        int need = 0;
        if (c < 0x80) { out.push_back(c); ++i; continue; }
        else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; need = 1; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; need = 2; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; need = 3; }
        else { out.push_back(0xFFFD); ++i; continue; }
        if (i + need >= len) {  // truncated sequence at the end of the text
            out.push_back(0xFFFD);
            break;
        }
        bool ok = true;
        for (int k = 1; k <= need; ++k) {
            const unsigned char cc = (unsigned char)p[i + k];
            if ((cc & 0xC0) != 0x80) { ok = false; break; }
            cp = (cp << 6) | (cc & 0x3F);
        }
        if (!ok) { out.push_back(0xFFFD); ++i; continue; }
        out.push_back(cp);
        i += need + 1;
    }
}

}  // namespace

struct rass_tokenizer {
    std::unordered_map<std::string, int32_t> vocab;
    int32_t unk = -1, cls = -1, sep = -1;
    bool lower = true;
    int max_chars = 100;

    void wordpiece(const std::vector<uint32_t>& word, std::vector<int32_t>& ids) const {
        if ((int)word.size() > max_chars) {
            ids.push_back(unk);
            return;
        }
        // byte offsets of code-point boundaries
        std::string bytes;
        std::vector<int> off(word.size() + 1, 0);
        for (size_t i = 0; i < word.size(); ++i) {
            utf8_append(bytes, word[i]);
            off[i + 1] = (int)bytes.size();
        }
        const size_t first = ids.size();
        size_t start = 0;
        std::string piece;
        while (start < word.size()) {
            size_t end = word.size();
            int32_t cur = -1;
            while (start < end) {
                piece.assign(start > 0 ? "##" : "");
                piece.append(bytes, (size_t)off[start], (size_t)(off[end] - off[start]));
                auto it = vocab.find(piece);
                if (it != vocab.end()) {
                    cur = it->second;
                    break;
                }
                --end;
            }
            if (cur < 0) {
                ids.resize(first);
                ids.push_back(unk);
                return;
            }
            ids.push_back(cur);
            start = end;
        }
    }

    // Unicode Final_Sigma as str.lower() applies it inside one whitespace-delimited token of the
    // CLEANED text: preceded by a cased letter (skipping case-ignorables), not followed by one.
    static bool token_break(uint32_t cp) { return (cp_flags(cp) & kSpace) || is_cjk(cp); }
    static bool removed(uint32_t cp) { return cp == 0 || cp == 0xFFFD || (cp_flags(cp) & kControl); }
    static bool final_sigma(const std::vector<uint32_t>& cps, size_t i) {
        bool before = false;
        for (size_t j = i; j-- > 0;) {
            const uint32_t c = cps[j];
            if (removed(c)) continue;
            if (token_break(c)) break;
            const uint8_t fl = cp_flags(c);
            if (fl & kCased) { before = true; break; }
            if (!(fl & kIgnorable)) break;
        }
        if (!before) return false;
        for (size_t j = i + 1; j < cps.size(); ++j) {
            const uint32_t c = cps[j];
            if (removed(c)) continue;
            if (token_break(c)) break;
            const uint8_t fl = cp_flags(c);
            if (fl & kCased) return false;
            if (!(fl & kIgnorable)) break;
        }
        return true;
    }

    int encode(const char* text, int64_t len, int max_len, int32_t* out) const {
        std::vector<uint32_t> cps;
        utf8_decode(text, len, cps);
        std::vector<int32_t> ids;
        ids.reserve(64);
        std::vector<uint32_t> word;
        const int budget = max_len - 2;
        auto flush = [&]() {
            if (!word.empty()) {
                wordpiece(word, ids);
                word.clear();
            }
        };
        for (size_t i = 0; i < cps.size() && (int)ids.size() < budget; ++i) {
            const uint32_t cp = cps[i];
            if (cp == 0 || cp == 0xFFFD) continue;
            const uint8_t fl = cp_flags(cp);
            if (fl & kSpace) { flush(); continue; }
            if (fl & kControl) continue;
            if (is_cjk(cp)) {  // every CJK character is its own word
                flush();
                word.push_back(cp);
                flush();
                continue;
            }
            uint32_t f[3];
            int n = lower ? cp_fold(cp, f) : (f[0] = cp, 1);
            if (lower && cp == 0x3A3 && final_sigma(cps, i)) {  // context-sensitive lower(): word-final sigma
                f[0] = 0x3C2;
                n = 1;
            }
            for (int k = 0; k < n; ++k) {
                if (cp_flags(f[k]) & kPunct) {  // punctuation splits and stands alone
                    flush();
                    word.push_back(f[k]);
                    flush();
                } else {
                    word.push_back(f[k]);
                }
            }
        }
        if ((int)ids.size() < budget) flush();
        int n = std::min<int>((int)ids.size(), budget);
        out[0] = cls;
        for (int i = 0; i < n; ++i) out[1 + i] = ids[(size_t)i];
        out[1 + n] = sep;
        return n + 2;
    }
};

extern "C" {

int rass_tokenizer_create(const char* vocab_path, int lower_case, rass_tokenizer_t** out) {
    if (!vocab_path || !out) return tfail(RASS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    std::ifstream f(vocab_path);
    if (!f) return tfail(RASS_ERR_IO, std::string("cannot open vocab: ") + vocab_path);
    rass_tokenizer* t = new (std::nothrow) rass_tokenizer();
    if (!t) return tfail(RASS_ERR_OOM, "host allocation failed");
    t->lower = lower_case != 0;
    std::string line;
    int32_t id = 0;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;  // the Python loader skips empty lines too
        t->vocab.emplace(line, id++);
    }
    auto get = [&](const char* k) {
        auto it = t->vocab.find(k);
        return it == t->vocab.end() ? -1 : it->second;
    };
    t->unk = get("[UNK]");
    t->cls = get("[CLS]");
    t->sep = get("[SEP]");
    if (t->unk < 0 || t->cls < 0 || t->sep < 0) {
        delete t;
        return tfail(RASS_ERR_INVALID, "vocab lacks [UNK] / [CLS] / [SEP]");
    }
    *out = t;
    return RASS_OK;
}

void rass_tokenizer_destroy(rass_tokenizer_t* t) { delete t; }

int rass_tokenizer_vocab_size(const rass_tokenizer_t* t) { return t ? (int)t->vocab.size() : 0; }

/* One text -> [CLS] pieces [SEP], at most max_len ids (max_len >= 2) into out_ids; returns the
 * count or a negative status. */
int rass_tokenizer_encode(const rass_tokenizer_t* t, const char* text, int64_t text_len, int max_len,
                          int32_t* out_ids) {
    if (!t || !out_ids || (!text && text_len > 0)) return tfail(RASS_ERR_INVALID, "NULL argument");
    if (max_len < 2 || text_len < 0) return tfail(RASS_ERR_INVALID, "max_len must be >= 2");
    return t->encode(text ? text : "", text_len, max_len, out_ids);
}

/* n texts -> packed ids (capacity n * max_len) + cu_seqlens[n + 1]; n_threads <= 0 = hardware
 * concurrency.  Returns the total token count or a negative status. */
int64_t rass_tokenizer_encode_batch(const rass_tokenizer_t* t, const char* const* texts, const int64_t* lens, int n,
                                    int max_len, int32_t* out_ids, int32_t* out_cu, int n_threads) {
    if (!t || !texts || !lens || !out_ids || !out_cu) return tfail(RASS_ERR_INVALID, "NULL argument");
    if (max_len < 2 || n < 0) return tfail(RASS_ERR_INVALID, "bad max_len / n");
    std::vector<int32_t> tmp((size_t)n * (size_t)max_len);
    std::vector<int32_t> cnt((size_t)n, 0);
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, std::max(1, n / 8)));
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i)
            cnt[(size_t)i] = t->encode(texts[i] ? texts[i] : "", lens[i], max_len, tmp.data() + (size_t)i * max_len);
    };
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; ++k) th.emplace_back(work, (int)((int64_t)n * k / nt), (int)((int64_t)n * (k + 1) / nt));
        for (auto& x : th) x.join();
    }
    int64_t total = 0;
    out_cu[0] = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(out_ids + total, tmp.data() + (size_t)i * max_len, (size_t)cnt[(size_t)i] * sizeof(int32_t));
        total += cnt[(size_t)i];
        out_cu[i + 1] = (int32_t)total;
    }
    return total;
}

}  // extern "C"
