// CPU sanitizer harness for the C++ WordPiece tokenizer (rassengine_amd/csrc/tokenizer.cpp), built by
// tests/test_tokenizer_sanitizers.py with -fsanitize=address,undefined: arbitrary bytes in (invalid and truncated UTF-8, NULs,
// overlong forms, surrogates, long runs without a space), every max_len edge, single and threaded batch entry points.  Any
// out-of-bounds access, use-after-free, signed overflow or misaligned access aborts the process with a report.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rass_engine.h"

extern "C" void rassint_set_last_error(const char*) {}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s vocab.txt\n", argv[0]); return 2; }
    rass_tokenizer_t* tok = nullptr;
    if (rass_tokenizer_create(argv[1], 1, &tok) != RASS_OK) { fprintf(stderr, "cannot load vocab\n"); return 2; }
    const char* pieces[] = {"the ", "patient", "'s ", "120/80", " ", "\t", "\n", "\xC3\xA9", "\xC3", "\xE4\xB8\xAD", "\xE4\xB8", "\xF0\x9F\x98\x80",
                            "\xF0\x9F", "\xED\xA0\x80", "\xC0\xAF", "\xFF", "\xFE", "\x00", "\x7F", "\xE2\x80\x8B", "\xCC\x81", "a", "Z", "##", "[UNK]",
                            "\xEF\xBF\xBD", "\xD0\xBF\xD1\x80\xD0\xB8", "\xCE\xA3", "\xC4\xB0", "\xE1\x84\x92\xE1\x85\xA1\xE1\x86\xAB"};
    const int n_pieces = (int)(sizeof(pieces) / sizeof(pieces[0]));
    const int max_lens[] = {2, 3, 4, 16, 48, 512};
    std::vector<std::string> texts;
    for (int it = 0; it < 3000; ++it) {
        std::string s;
        const int parts = (int)(rnd() % 40);
        for (int p = 0; p < parts; ++p) {
            const uint32_t r = rnd();
            if (r % 7 == 0) {                                  // raw random bytes
                const int nb = 1 + (int)(rnd() % 6);
                for (int b = 0; b < nb; ++b) s.push_back((char)(rnd() & 0xff));
            } else if (r % 31 == 1) {                          // a long run without a space (the > 100 characters rule)
                s.append(90 + rnd() % 40, (char)('a' + rnd() % 26));
            } else {
                const char* pc = pieces[r % n_pieces];
                s.append(pc, pc[0] == 0 ? 1 : strlen(pc));    // the NUL piece is one byte
            }
        }
        texts.push_back(s);
    }
    texts.push_back(std::string());
    texts.push_back(std::string(5000, 'x'));
    texts.push_back(std::string(3000, '\xE4'));
    long long total = 0;
    std::vector<int32_t> ids(512 + 8);
    for (const std::string& s : texts)
        for (int ml : max_lens) {
            // the output buffer is EXACTLY max_len long: a write past it is an ASAN report
            std::vector<int32_t> out((size_t)ml);
            const int n = rass_tokenizer_encode(tok, s.data(), (int64_t)s.size(), ml, out.data());
            if (n < 2 || n > ml) { fprintf(stderr, "bad count %d for max_len %d\n", n, ml); return 1; }
            total += n;
        }
    // batch entry point, threaded
    for (int ml : {2, 16, 512}) {
        std::vector<const char*> ptrs; std::vector<int64_t> lens;
        for (const std::string& s : texts) { ptrs.push_back(s.data()); lens.push_back((int64_t)s.size()); }
        ptrs[3] = nullptr; lens[3] = 0;                        // a NULL text of length 0 is an empty text
        std::vector<int32_t> out(texts.size() * (size_t)ml), cu(texts.size() + 1);
        const int64_t n = rass_tokenizer_encode_batch(tok, ptrs.data(), lens.data(), (int)texts.size(), ml, out.data(), cu.data(), 4);
        if (n < 2 * (int64_t)texts.size() || cu[texts.size()] != n) { fprintf(stderr, "bad batch total\n"); return 1; }
        total += n;
    }
    // argument errors come back as statuses, not crashes
    if (rass_tokenizer_encode(tok, "x", 1, 1, ids.data()) >= 0) return 1;
    if (rass_tokenizer_encode(nullptr, "x", 1, 8, ids.data()) >= 0) return 1;
    if (rass_tokenizer_encode(tok, nullptr, 3, 8, ids.data()) >= 0) return 1;
    rass_tokenizer_destroy(tok);
    printf("ok %lld tokens\n", total);
    return 0;
}
