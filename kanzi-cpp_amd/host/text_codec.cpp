// Host-side stages of the reference's level presets 5 and 6: the TEXT transform (both word-index encodings) and the UTF transform.
// They run on the host, in front of the device chain (BWT + RANK / SRT + ZRLT and the entropy coder): a word-replacement pass over a
// hash map and an alias table over code points are short dependent chains per block with data-dependent output, i.e. control plane
// next to the suffix sort -- but the bytes they produce decide the rest of the stream, so they are restated to the letter.
//
// Reference being replaced (bit-identical output, same accept / refuse decisions):
//   transform/TextCodec.cpp:120-211  character classes, static dictionary (the 1,024 words of text_words_en.inc)
//   transform/TextCodec.cpp:213-425  block statistics: text / XML / CRLF flags, data type of blocks that are not text
//   transform/TextCodec.cpp:520-1015 word indexes behind an escape byte ("codec 1": entropy FPAQ / CM / TPAQ)
//   transform/TextCodec.cpp:1017-1581 word indexes with the top bit set ("codec 2": entropy NONE / ANS0 / HUFFMAN / RANGE)
//   transform/TransformFactory.hpp:225-242 which of the two a stream uses
//   transform/UTFCodec.cpp:47-420     UTF-8 code points replaced by one- or two-byte aliases in order of frequency
//   Global.cpp:354-397 (detectSimpleType), Magic.hpp:64-170 (magic numbers), io/CompressedOutputStream.cpp:724-733 (data type preset)
#include "host_stages.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

namespace kanzi_amd {
namespace hoststage {

// ------------------------------------------------------------------------------------------------
// magic numbers and the data type a block starts with
// ------------------------------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

uint32_t magicOf(const uint8_t* p)
{
    static const uint32_t four[] = { 0x47494638u /* GIF */, 0x25504446u /* PDF */, 0x504B0304u /* ZIP */, 0x377ABCAFu /* 7z */, 0x89504E47u /* PNG */,
                                     0x7F454C46u /* ELF */, 0xFEEDFACEu, 0xCEFAEDFEu, 0xFEEDFACFu, 0xCFFAEDFEu /* Mach-O */, 0x28B52FFDu /* zstd */,
                                     0x81CFB2CEu /* brotli */, 0x4D534346u /* CAB */, 0x52494646u /* RIFF */, 0x664C6143u /* FLAC */, 0xFD377A58u /* xz */,
                                     0x4B414E5Au /* KANZ */, 0x52617221u /* RAR */ };
    const uint32_t k = be32(p);
    if ((k & ~0x0Fu) == 0xFFD8FFE0u) return k;                                  // JPEG (the low nibble stays in the value)
    if ((k >> 8) == 0x425A68u || (k >> 8) == 0x494433u) return k >> 8;          // bzip2, ID3
    for (uint32_t m : four) if (k == m) return k;
    const uint32_t k16 = k >> 16;
    if (k16 == 0x1F8Bu || k16 == 0x424Du || k16 == 0x4D5Au) return k16;         // gzip, BMP, MZ
    if (k16 == 0x5034u || k16 == 0x5035u || k16 == 0x5036u) {                   // binary PBM / PGM / PPM: "P4".."P6" + white space
        const uint32_t c = (k >> 8) & 0xFF;
        if (c == 0x07 || c == 0x0A || c == 0x0D || c == 0x20) return k16;
    }
    return 0;
}

static bool magicCompressed(uint32_t m)
{
    switch (m) {
    case 0xFFD8FFE0u: case 0x47494638u: case 0x89504E47u: case 0x377ABCAFu: case 0x28B52FFDu: case 0x81CFB2CEu: case 0x4D534346u: case 0x504B0304u:
    case 0x1F8Bu: case 0x425A68u: case 0x664C6143u: case 0x494433u: case 0xFD377A58u: case 0x4B414E5Au: case 0x52617221u:
        return true;
    default:
        return false;
    }
}
static bool magicMultimedia(uint32_t m)
{
    switch (m) {
    case 0xFFD8FFE0u: case 0x47494638u: case 0x89504E47u: case 0x52494646u: case 0x664C6143u: case 0x494433u: case 0x424Du: case 0x5034u: case 0x5035u: case 0x5036u:
        return true;
    default:
        return false;
    }
}
static bool magicExecutable(uint32_t m)
{
    switch (m) {
    case 0x7F454C46u: case 0x4D5Au: case 0xFEEDFACEu: case 0xCEFAEDFEu: case 0xFEEDFACFu: case 0xCFFAEDFEu:
        return true;
    default:
        return false;
    }
}

int presetDataType(const uint8_t* block, int n)
{
    if (n < 4) return DT_UNDEFINED;
    const uint32_t m = magicOf(block);
    if (magicCompressed(m)) return DT_BIN;
    if (magicMultimedia(m)) return DT_MULTIMEDIA;
    if (magicExecutable(m)) return DT_EXE;
    return DT_UNDEFINED;
}

// ------------------------------------------------------------------------------------------------
// character classes: 0 letter, 1 delimiter, -1 anything else
// ------------------------------------------------------------------------------------------------
static const int8_t* charClasses()
{
    static int8_t tab[256];
    static bool ready = false;
    if (!ready) {
        for (int c = 0; c < 256; c++) {
            int8_t t = -1;
            if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) t = 0;
            if ((c >= ' ' && c <= '/') || (c >= ':' && c <= '?')) t = 1;
            if (c == '\n' || c == '\r' || c == '\t' || c == '_' || c == '|' || c == '{' || c == '}' || c == '[' || c == ']') t = 1;
            tab[c] = t;
        }
        ready = true;
    }
    return tab;
}
static inline bool isLetter(uint8_t c) { return charClasses()[c] == 0; }

// ------------------------------------------------------------------------------------------------
// block statistics
// ------------------------------------------------------------------------------------------------
enum : uint8_t { F_NOT_TEXT = 0x80, F_CRLF = 0x40, F_XML = 0x20, F_CODEC2 = 0x10, F_TYPE = 0x0F };

struct PairCounts {
    std::vector<uint32_t> f1;          // 65,536 counts of (previous byte, byte), the byte in front of the block taken as 0
    uint32_t f0[256];
    PairCounts(const uint8_t* p, int n) : f1(65536, 0u)
    {
        memset(f0, 0, sizeof(f0));
        uint32_t prev = 0;
        for (int i = 0; i < n; i++) { const uint32_t c = p[i]; f0[c]++; f1[(prev << 8) | c]++; prev = c; }
    }
};

// what the pair statistics say about UTF-8: no byte or pair that cannot occur, and at least an eighth of continuation bytes
static bool looksLikeUtf8(const uint32_t* f0, const std::vector<uint32_t>& f1, int n)
{
    uint32_t bad = f0[0xC0] + f0[0xC1];
    for (int c = 0xF5; c <= 0xFF; c++) bad += f0[c];
    if (bad) return false;
    uint32_t cont = 0;
    for (int c = 0; c < 256; c++) {
        if (c < 0xA0 || c > 0xBF) bad += f1[0xE0 * 256 + c];
        if (c < 0x80 || c > 0x9F) bad += f1[0xED * 256 + c];
        if (c < 0x90 || c > 0xBF) bad += f1[0xF0 * 256 + c];
        if (c < 0x80 || c > 0x8F) bad += f1[0xF4 * 256 + c];
        if (c < 0x80 || c > 0xBF) {
            for (int l = 0xC2; l <= 0xDF; l++) bad += f1[l * 256 + c];
            for (int l = 0xE1; l <= 0xEC; l++) bad += f1[l * 256 + c];
            for (int l = 0xEE; l <= 0xF3; l++) bad += f1[l * 256 + c];
        } else {
            cont += f0[c];
        }
        if (bad) return false;
    }
    return cont >= uint32_t(n / 8);
}

static int simpleType(int n, const uint32_t* f0)
{
    int sum = 0;
    for (const char* s = "acgntuACGNTU"; *s; s++) sum += int(f0[uint8_t(*s)]);
    if (sum > n - n / 12) return DT_DNA;
    sum = 0;
    for (const char* s = "0123456789+-*/=,.:; "; *s; s++) sum += int(f0[uint8_t(*s)]);
    if (sum == n) return DT_NUMERIC;
    sum = (f0['='] == 1) ? 1 : 0;
    for (const char* s = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"; *s; s++) sum += int(f0[uint8_t(*s)]);
    if (sum == n) return DT_BASE64;
    int distinct = 0;
    for (int c = 0; c < 256; c++) distinct += f0[c] ? 1 : 0;
    if (distinct == 256) return DT_BIN;
    return distinct <= 4 ? DT_SMALL_ALPHABET : DT_UNDEFINED;
}

// flags of a block: F_NOT_TEXT | type, or F_CRLF / F_XML for text. `strict` is codec 1's test, the other one is codec 2's.
static uint8_t blockFlags(const uint8_t* p, int n, bool strict)
{
    if (!strict && magicOf(p) != 0) return F_NOT_TEXT;
    const PairCounts pc(p, n);
    const uint32_t* f0 = pc.f0;
    int letters = int(f0['\r'] + f0['\n']), ascii = 0;
    for (int c = 0; c < 128; c++) { if (isLetter(uint8_t(c))) letters += int(f0[c]); ascii += int(f0[c]); }
    const int high = n - ascii;
    bool notText = high > (n >> 2);
    if (!notText) {
        notText = letters < (n >> 2);
        if (strict) notText |= (f0[0] >= uint32_t(n / 100)) || ((ascii / 95) < (n / 100));
        else notText |= f0[' '] < uint32_t(n / 50);
    }
    if (notText) {
        const int t = simpleType(n, f0);
        if (t != DT_UNDEFINED) return uint8_t(F_NOT_TEXT | t);
        return looksLikeUtf8(f0, pc.f1, n) ? uint8_t(F_NOT_TEXT | DT_UTF8) : uint8_t(F_NOT_TEXT);
    }
    uint8_t res = 0;
    if (high <= n - n / 10) {
        // '<' and '>' about equally often, often enough, and some "&a", "&g", "&l", "&q"
        const int lt = int(f0['<']), gt = int(f0['>']);
        const int amp = int(pc.f1['&' * 256 + 'a'] + pc.f1['&' * 256 + 'g'] + pc.f1['&' * 256 + 'l'] + pc.f1['&' * 256 + 'q']);
        const int least = std::max((n - high) >> 9, 2);
        if (lt >= least && gt >= least && amp > 0) {
            if (lt < gt) { if (lt >= gt - gt / 100) res |= F_XML; }
            else if (gt < lt) { if (gt >= lt - lt / 100) res |= F_XML; }
            else res |= F_XML;
        }
    }
    if (f0['\r'] != 0 && f0['\r'] == f0['\n']) {
        res |= F_CRLF;
        for (int c = 0; c < 256; c++) {
            if (c != '\n' && pc.f1['\r' * 256 + c] != 0) { res &= uint8_t(~F_CRLF); break; }
            if (c != '\r' && pc.f1[c * 256 + '\n'] != 0) { res &= uint8_t(~F_CRLF); break; }
        }
    }
    return res;
}

// ------------------------------------------------------------------------------------------------
// the dictionary: static words, then the words of the block in order of first appearance
// ------------------------------------------------------------------------------------------------
static const uint32_t H1 = 0x7FEB352Du, H2 = 0x846CA68Bu;
static const int MAX_WORD = 31, MAX_ENTRIES = 1 << 19, LEN_MASK = 0x0007FFFF;
static const int T1 = 128, T2 = T1 * T1, T3 = 64, T4 = T3 * 128;
static const uint8_t ESC1 = 0x0F, ESC2 = 0x0E, CR = 0x0D, LF = 0x0A, SP = 0x20;

static inline uint32_t hashStep(uint32_t h, uint32_t c) { return (h * H1) ^ (c * H2); }
static uint32_t wordHash(const uint8_t* w, int len) { uint32_t h = H1; for (int i = 0; i < len; i++) h = hashStep(h, w[i]); return h; }

struct Word { const uint8_t* text; uint32_t hash; int32_t lenIdx; };       // lenIdx = length << 24 | index

struct StaticWords {
    std::vector<uint8_t> chars;
    std::vector<Word> words;
    StaticWords()
    {
        static const char list[] =
#include "text_words_en.inc"
            ;
        chars.assign(list, list + sizeof(list) - 1);
        size_t at = 0;
        while (at < chars.size()) {
            size_t e = at;
            while (e < chars.size() && chars[e] != ' ') e++;
            if (e > at) { const int len = int(e - at); words.push_back(Word{ &chars[at], wordHash(&chars[at], len), int32_t((len << 24) | int(words.size())) }); }
            at = e + 1;
        }
    }
};
static const StaticWords& staticWords() { static const StaticWords s; return s; }

class Dictionary {
public:
    std::vector<Word> list;
    std::vector<int32_t> slot;          // hash & mask -> index into list, -1 = empty
    uint32_t mask;
    int fixed;                          // entries that are never replaced
    uint8_t escapes[2];

    // codec 1 appends two one-byte entries (the escape bytes themselves) to the static words
    Dictionary(int logSlots, int count, bool withEscapes)
    {
        mask = (1u << logSlots) - 1u;
        slot.assign(size_t(1) << logSlots, -1);
        const StaticWords& sw = staticWords();
        const int nStatic = int(sw.words.size());
        int lg = 13;
        if (count >= 1024) { lg = 31 - __builtin_clz(uint32_t(count / 128)); lg = std::max(std::min(lg, 18), 13); }
        const int size = std::max(nStatic + (withEscapes ? 2 : 0), 1 << lg);
        list.assign(sw.words.begin(), sw.words.end());
        fixed = nStatic;
        if (withEscapes) {
            escapes[0] = ESC2; escapes[1] = ESC1;
            list.push_back(Word{ &escapes[0], 0u, int32_t((1 << 24) | fixed) });
            list.push_back(Word{ &escapes[1], 0u, int32_t((1 << 24) | (fixed + 1)) });
            fixed += 2;
        }
        for (int i = 0; i < fixed; i++) slot[list[size_t(i)].hash & mask] = i;
        for (int i = fixed; i < size; i++) list.push_back(Word{ nullptr, 0u, int32_t(i) });
    }
    int size() const { return int(list.size()); }
    bool grow()
    {
        const int n = size();
        if (n >= MAX_ENTRIES) return false;
        for (int i = n; i < 2 * n; i++) list.push_back(Word{ nullptr, 0u, int32_t(i) });
        for (int i = 0; i < n; i++) slot[list[size_t(i)].hash & mask] = i;      // (every old entry claims its slot again, unused ones slot 0)
        return true;
    }
    // the word [w, w + len) with hash h takes entry `at` (an entry is reused once the index has wrapped)
    void put(int& at, const uint8_t* w, int len, uint32_t h)
    {
        Word& e = list[size_t(at)];
        if ((e.lenIdx & LEN_MASK) >= fixed) {
            slot[e.hash & mask] = -1;
            e.text = w; e.hash = h; e.lenIdx = int32_t((len << 24) | at);
        }
        slot[h & mask] = at;
        at++;
        if (at >= size() && !grow()) at = fixed;
    }
};

static bool sameTail(const uint8_t* a, const uint8_t* b, int len) { return len <= 0 || memcmp(a, b, size_t(len)) == 0; }

// ------------------------------------------------------------------------------------------------
// the two encodings
// ------------------------------------------------------------------------------------------------
struct Enc1 {
    static const bool escapes = true, strict = true;
    static int slack() { return 4; }
    static int logSlots(int blockSize) { return blockSize >= 8 ? std::max(std::min(31 - __builtin_clz(uint32_t(blockSize / 8)), 26), 13) : 13; }
    static int index(uint8_t* d, int v)
    {
        if (v >= T1) {
            if (v >= T2) { d[0] = uint8_t(0xE0 | (v >> 14)); d[1] = uint8_t(0x80 | (v >> 7)); d[2] = uint8_t(0x7F & v); return 3; }
            d[0] = uint8_t(0x80 | (v >> 7)); d[1] = uint8_t(0x7F & v); return 2;
        }
        d[0] = uint8_t(v); return 1;
    }
    static int reference(uint8_t* d, int v, bool flipped) { d[0] = flipped ? ESC2 : ESC1; return 1 + index(d + 1, v); }
    // literal bytes; -1 when they do not fit
    static int literals(const uint8_t* s, uint8_t* d, int n, int room, bool crlf, int fixed)
    {
        int o = 0;
        for (int i = 0; i < n; i++) {
            if (o >= room) return -1;
            const uint8_t c = s[i];
            if (c == ESC1 || c == ESC2) {
                d[o++] = ESC1;
                const int v = (c == ESC1) ? fixed - 1 : fixed - 2;
                const int need = v >= T1 ? (v >= T2 ? 3 : 2) : 1;
                if (o + need >= room) return -1;
                o += index(d + o, v);
            } else if (c == CR) {
                if (!crlf) d[o++] = c;
            } else {
                d[o++] = c;
            }
        }
        return o;
    }
};

struct Enc2 {
    static const bool escapes = false, strict = false;
    static int slack() { return 3; }
    static int logSlots(int blockSize) { return blockSize >= 32 ? std::max(std::min(31 - __builtin_clz(uint32_t(blockSize / 32)), 24), 13) : 13; }
    static int index(uint8_t* d, int v)
    {
        v++;                                                             // 0x80 alone means "first letter's case flipped"
        if (v >= T3) {
            if (v >= T4) { d[0] = uint8_t(0xF0 | (v >> 16)); d[1] = uint8_t(v >> 8); d[2] = uint8_t(v); return 3; }
            d[0] = uint8_t(0xC0 | (v >> 8)); d[1] = uint8_t(v); return 2;
        }
        d[0] = uint8_t(0x80 | v); return 1;
    }
    static int reference(uint8_t* d, int v, bool flipped) { d[0] = 0x80; const int o = flipped ? 1 : 0; return o + index(d + o, v); }
    static int literals(const uint8_t* s, uint8_t* d, int n, int room, bool crlf, int)
    {
        int o = 0;
        const bool checked = !(2 * n < room);
        for (int i = 0; i < n; i++) {
            const uint8_t c = s[i];
            if (c == ESC1) {
                if (checked && o >= room - 1) return -1;
                d[o++] = ESC1; d[o++] = ESC1;
            } else if (c == CR) {
                if (!crlf) { if (checked && o >= room) return -1; d[o++] = c; }
            } else {
                if (c >= 128) { if (checked && o >= room) return -1; d[o++] = ESC1; }
                if (checked && o >= room) return -1;
                d[o++] = c;
            }
        }
        return o;
    }
};

template <class E>
static bool wordsForward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int& dataType, int& outLen)
{
    outLen = 0;
    if (dstCap < count) return false;
    if (dataType != DT_UNDEFINED && dataType != DT_TEXT && dataType != DT_BIN) return false;
    const uint8_t flags = blockFlags(src, count, E::strict);
    if (flags & F_NOT_TEXT) { dataType = flags & F_TYPE; return false; }
    dataType = DT_TEXT;
    Dictionary dict(E::logSlots(blockSize), count, E::escapes);
    const int8_t* cls = charClasses();
    const int end = count, room = count, roomRef = room - E::slack();
    const bool crlf = (flags & F_CRLF) != 0;
    int at = dict.fixed, emitted = 0, i = 0, o = 0;
    dst[o++] = flags;
    while (i < end && src[i] == SP) { dst[o++] = SP; i++; emitted++; }
    int delim = (i < end && isLetter(src[i])) ? i - 1 : i;               // the delimiter in front of the word being read
    uint32_t h = H1, hFlip = H1;
    bool ok = true;
    while (i < end) {
        const uint8_t c = src[i];
        const int8_t t = cls[c];
        if (t == 0) {
            if (i - delim == 1) { h = hashStep(H1, c); hFlip = hashStep(H1, uint32_t(c) ^ 0x20u); }
            else { h = hashStep(h, c); hFlip = hashStep(hFlip, c); }
            i++;
            continue;
        }
        if (i > delim + 2 && t > 0) {
            const int len = i - delim - 1;
            if (len <= MAX_WORD) {
                const int s1 = dict.slot[h & dict.mask];
                int hit = -1;
                if (s1 >= 0 && dict.list[size_t(s1)].hash == h && (dict.list[size_t(s1)].lenIdx >> 24) == len) hit = s1;
                else {
                    const int s2 = dict.slot[hFlip & dict.mask];
                    if (s2 >= 0 && dict.list[size_t(s2)].hash == hFlip && (dict.list[size_t(s2)].lenIdx >> 24) == len) hit = s2;
                }
                if (hit >= 0 && !sameTail(dict.list[size_t(hit)].text + 1, src + delim + 2, len - 1)) hit = -1;
                if (hit < 0) {
                    if ((len > 3 || (len == 3 && at < T2)) && s1 < 0) dict.put(at, src + delim + 1, len, h);
                } else {
                    // a single space between two word references is implied
                    if (emitted != delim || src[delim] != SP) {
                        const int k = E::literals(src + emitted, dst + o, delim + 1 - emitted, room - o, crlf, dict.fixed);
                        if (k < 0) { ok = false; break; }
                        o += k;
                    }
                    if (o >= roomRef) { ok = false; break; }
                    const Word& w = dict.list[size_t(hit)];
                    o += E::reference(dst + o, w.lenIdx & LEN_MASK, hit != s1);
                    emitted = delim + 1 + (w.lenIdx >> 24);
                }
            }
        }
        delim = i;
        i++;
    }
    if (ok) {
        const int k = E::literals(src + emitted, dst + o, end - emitted, room - o, crlf, dict.fixed);
        if (k < 0) ok = false; else o += k;
        ok = ok && (i == end);
    }
    outLen = o;
    return ok;
}

template <class E>
static bool wordsInverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int& outLen)
{
    outLen = 0;
    if (count < 2) return false;
    Dictionary dict(E::logSlots(blockSize), dstCap, E::escapes);
    const int8_t* cls = charClasses();
    const bool crlf = (src[0] & F_CRLF) != 0;
    const bool oldIndexes = !E::escapes && bsVersion < 6;
    const int end = count, room = dstCap;
    int i = 1, o = 0, at = dict.fixed;
    int delim = isLetter(src[i]) ? i - 1 : i;
    bool afterWord = false, ok = true;
    // the bytes of a word index (and the literal behind an escape) may be cut off by the end of a damaged block: nothing is
    // read behind src[end - 1] (the caller's slice has no padding); a block that ends inside an index fails the final i == end
    auto rd = [src, end](int k) -> int { return k < end ? int(src[k]) : 0; };
    while (i < end && o < room) {
        uint8_t c = src[i];
        const int8_t t = cls[c];
        if (t == 0) { dst[o++] = src[i++]; continue; }
        if (i > delim + 3 && t > 0) {
            const int len = i - delim - 1;
            if (len <= MAX_WORD) {
                const uint32_t h = wordHash(src + delim + 1, len);
                const int s1 = dict.slot[h & dict.mask];
                bool known = false;
                if (s1 >= 0 && dict.list[size_t(s1)].hash == h && (dict.list[size_t(s1)].lenIdx >> 24) == len)
                    known = sameTail(dict.list[size_t(s1)].text + 1, src + delim + 2, len - 1);
                if (!known && (len > 3 || at < T2) && s1 < 0) dict.put(at, src + delim + 1, len, h);
            }
        }
        i++;
        bool isRef;
        uint8_t flip = 0;
        int idx = 0;
        if (E::escapes) {
            isRef = (c == ESC1 || c == ESC2);
            if (isRef) {
                idx = rd(i++);
                if (idx >= 128) {
                    const int b2 = rd(i++);
                    if (b2 >= 128) { idx = ((idx & 0x1F) << 14) | ((b2 & 0x7F) << 7) | rd(i); i++; }
                    else idx = ((idx & 0x7F) << 7) | b2;
                    if (idx >= dict.size()) { ok = false; break; }
                }
                if (c == ESC2) flip = 0x20;
            }
        } else {
            isRef = c >= 0x80;
            if (isRef) {
                if (oldIndexes) {
                    flip = c & 0x20;
                    idx = c & 0x1F;
                    if (c & 0x40) {
                        const int b2 = rd(i++);
                        if (b2 >= 128) { idx = (idx << 14) | ((b2 & 0x7F) << 7) | rd(i); i++; }
                        else idx = (idx << 7) | b2;
                        if (idx >= dict.size()) { ok = false; break; }
                    }
                } else {
                    if (c == 0x80) { flip = 0x20; c = uint8_t(rd(i++)); }
                    idx = c & 0x7F;
                    if (idx >= 64) {
                        if (idx >= 112) { idx = ((idx & 0x0F) << 16) | (rd(i) << 8) | rd(i + 1); i += 2; }
                        else { idx = ((idx & 0x1F) << 8) | rd(i); i++; }
                        if (idx > dict.size()) { ok = false; break; }
                    }
                    // an index of 0 in any of its encodings names no word: the reference checks the one-byte form only
                    // (TextCodec.cpp:1507-1518) and reads _dictList[-1] for C0 00 / F0 00 00; refused here
                    if (idx == 0) { ok = false; break; }
                    idx--;
                }
            }
        }
        if (isRef) {
            const Word& w = dict.list[size_t(idx)];
            const int len = (w.lenIdx >> 24) & 0xFF;
            if (len > 1) {
                if (afterWord) dst[o++] = SP;
                afterWord = true;
                delim = i;
            } else {
                if (len == 0) { ok = false; break; }
                afterWord = false;
                delim = i - 1;
            }
            if (o + len > room) { ok = false; break; }
            memcpy(dst + o, w.text, size_t(len));
            dst[o] ^= flip;
            o += len;
        } else {
            if (!E::escapes && c == ESC1) {
                dst[o++] = uint8_t(rd(i++));
            } else {
                if (crlf && c == LF) {
                    dst[o++] = CR;
                    if (o >= room) { ok = false; break; }
                }
                dst[o++] = c;
            }
            afterWord = false;
            delim = i - 1;
        }
    }
    outLen = o;
    return ok && i == end;
}

bool textForward(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* dataType, int* outLen)
{
    *outLen = 0;
    if (count == 0) return true;
    if (count < 1024 || count > (1 << 30)) return false;
    const bool ok = (variant == 1) ? wordsForward<Enc1>(src, count, dst, dstCap, blockSize, *dataType, *outLen)
                                   : wordsForward<Enc2>(src, count, dst, dstCap, blockSize, *dataType, *outLen);
    if (ok && bsVersion >= 7) { if (variant == 1) dst[0] &= uint8_t(~F_CODEC2); else dst[0] |= F_CODEC2; }
    return ok;
}

bool textInverse(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* outLen)
{
    *outLen = 0;
    if (count == 0) return true;
    if (count > (1 << 30) || count < 2) return false;
    if (bsVersion >= 7) variant = (src[0] & F_CODEC2) ? 2 : 1;
    return (variant == 1) ? wordsInverse<Enc1>(src, count, dst, dstCap, blockSize, bsVersion, *outLen)
                          : wordsInverse<Enc2>(src, count, dst, dstCap, blockSize, bsVersion, *outLen);
}

int textVariantFor(const char* entropy)
{
    std::string e(entropy ? entropy : "");
    for (char& ch : e) ch = char(toupper(ch));
    return (e == "NONE" || e == "ANS0" || e == "HUFFMAN" || e == "RANGE") ? 2 : 1;
}

// ------------------------------------------------------------------------------------------------
// UTF
// ------------------------------------------------------------------------------------------------
static const uint8_t* utfLengths()
{
    static uint8_t tab[256];
    static bool ready = false;
    if (!ready) {
        for (int c = 0; c < 256; c++) tab[c] = c < 0x80 ? 1 : (c < 0xC2 ? 0 : (c < 0xE0 ? 2 : (c < 0xF0 ? 3 : (c < 0xF5 ? 4 : 0))));
        ready = true;
    }
    return tab;
}

// the sequence at p as a 22-bit number: 3 size bits, then the byte itself, the two raw bytes, or the code point's payload bits; its
// length comes from the lead byte's top nibble alone (0 = a continuation byte where a sequence should start)
static int utfPack(const uint8_t* p, uint32_t& out)
{
    switch (p[0] >> 4) {
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 7:
        out = p[0]; return 1;
    case 12: case 13:
        out = (1u << 19) | (uint32_t(p[0]) << 8) | uint32_t(p[1]); return 2;
    case 14:
        out = (2u << 19) | (uint32_t(p[0] & 0x0F) << 12) | (uint32_t(p[1] & 0x3F) << 6) | uint32_t(p[2] & 0x3F); return 3;
    case 15:
        out = (4u << 19) | (uint32_t(p[0] & 0x07) << 18) | (uint32_t(p[1] & 0x3F) << 12) | (uint32_t(p[2] & 0x3F) << 6) | uint32_t(p[3] & 0x3F); return 4;
    default:
        out = 0; return 0;
    }
}

static int utfUnpack(uint32_t v, uint8_t* d)
{
    switch (v >> 19) {
    case 0: d[0] = uint8_t(v); return 1;
    case 1: d[0] = uint8_t(v >> 8); d[1] = uint8_t(v); return 2;
    case 2: d[0] = uint8_t(0xE0 | ((v >> 12) & 0x0F)); d[1] = uint8_t(0x80 | ((v >> 6) & 0x3F)); d[2] = uint8_t(0x80 | (v & 0x3F)); return 3;
    case 4: case 5: case 6: case 7:
        d[0] = uint8_t(0xF0 | ((v >> 18) & 0x07)); d[1] = uint8_t(0x80 | ((v >> 12) & 0x3F)); d[2] = uint8_t(0x80 | ((v >> 6) & 0x3F)); d[3] = uint8_t(0x80 | (v & 0x3F)); return 4;
    default: return 0;
    }
}

bool utfForward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* dataType, int* outLen)
{
    *outLen = 0;
    if (count == 0) return true;
    if (count < 1024) return false;
    if (dstCap < count) return false;
    if (*dataType != DT_UNDEFINED && *dataType != DT_UTF8) return false;
    const bool validate = *dataType != DT_UTF8;
    const uint8_t* lens = utfLengths();
    int start = 0;
    if (src[0] == 0xEF && src[1] == 0xBB && src[2] == 0xBF) start = 3;
    else while (start < 4 && lens[src[start]] == 0) start++;
    if (validate) {
        const int m = count - start - 4;
        const PairCounts pc(src + start, m);
        if (!looksLikeUtf8(pc.f0, pc.f1, m)) return false;
    }
    *dataType = DT_UTF8;
    std::vector<uint32_t> alias(size_t(1) << 22, 0u);
    struct Sym { uint32_t val, freq; };
    std::vector<Sym> syms;
    int n = 0;
    bool ok = true;
    for (int i = start; i < count - 4;) {
        uint32_t v;
        const int s = utfPack(src + i, v);
        ok = s != 0;
        ok = ok && (s != 3 || (src[i + 2] & 0xC0) == 0x80);
        ok = ok && (s != 4 || ((((uint32_t(src[i + 2]) << 8) | src[i + 3]) & 0xC0C0u) == 0x8080u));
        if (alias[v] == 0) { n++; ok = ok && n < 32768; syms.push_back(Sym{ v, 0u }); }
        if (!ok) break;
        alias[v]++;
        i += s;
    }
    const int limit = count - count / 10;
    if (!ok || n == 0 || 3 * n + 6 >= limit) return false;
    for (Sym& s : syms) s.freq = alias[s.val];
    // by decreasing frequency, then by decreasing value
    std::sort(syms.begin(), syms.end(), [](const Sym& a, const Sym& b) { return a.freq != b.freq ? a.freq > b.freq : a.val > b.val; });
    int o = 2;
    dst[o++] = uint8_t(n >> 8);
    dst[o++] = uint8_t(n);
    int estimate = o + 6;
    for (int k = 0; k < n; k++) {
        estimate += int(k < 128 ? syms[size_t(k)].freq : 2 * syms[size_t(k)].freq);
        const uint32_t v = syms[size_t(k)].val;
        alias[v] = (k < 128) ? uint32_t(k) : (0x10080u | ((uint32_t(k) << 1) & 0xFF00u) | (uint32_t(k) & 0x7Fu));
        dst[o] = uint8_t(v >> 16); dst[o + 1] = uint8_t(v >> 8); dst[o + 2] = uint8_t(v);
        o += 3;
    }
    if (estimate >= limit) return false;
    for (int i = 0; i < start; i++) dst[o++] = src[i];
    int i = start;
    while (i < count - 4) {
        uint32_t v;
        i += utfPack(src + i, v);
        const uint32_t a = alias[v];
        dst[o++] = uint8_t(a);
        dst[o] = uint8_t(a >> 8);
        o += int(a >> 16);
    }
    dst[0] = uint8_t(start);
    dst[1] = uint8_t(i - (count - 4));
    while (i < count) dst[o++] = src[i++];
    *outLen = o;
    return o < limit;
}

bool utfInverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* outLen)
{
    *outLen = 0;
    if (count == 0) return true;
    if (count < 4) return false;
    const int start = src[0] & 3, adjust = src[1] & 3;
    const int n = (int(src[2]) << 8) + int(src[3]);
    if (n == 0 || n >= 32768 || 3 * n > count - 4) return false;
    struct Sym { uint8_t bytes[4]; uint8_t len; };
    std::vector<Sym> m;
    m.resize(size_t(n));
    int i = 4;
    for (int k = 0; k < n; k++) {
        if (i + 3 > count) return false;
        const uint32_t v = (uint32_t(src[i]) << 16) | (uint32_t(src[i + 1]) << 8) | uint32_t(src[i + 2]);
        memset(m[size_t(k)].bytes, 0, 4);
        const int l = utfUnpack(v, m[size_t(k)].bytes);
        if (l == 0) return false;
        m[size_t(k)].len = uint8_t(l);
        i += 3;
    }
    int o = 0;
    const int srcEnd = count - 4 + adjust, dstEnd = dstCap - 4;
    if (dstEnd < 0) return false;
    if (srcEnd > count || i + start > srcEnd || o + start > dstCap) return false;
    for (int k = 0; k < start; k++) dst[o++] = src[i++];
    while (i < srcEnd) {
        uint32_t a = src[i++];
        if (a >= 128) a = (uint32_t(src[i++]) << 7) + (a & 0x7F);
        if (a >= uint32_t(n)) return false;
        const Sym& s = m[size_t(a)];
        if (o + int(s.len) > dstCap) return false;
        if (o + 4 <= dstCap) memcpy(dst + o, s.bytes, 4); else memcpy(dst + o, s.bytes, s.len);
        o += s.len;
    }
    if (i == srcEnd && o < dstEnd + adjust) {
        if (i + 4 - adjust > count || o + 4 - adjust > dstCap) return false;
        for (int k = 0; k < 4 - adjust; k++) dst[o++] = src[i++];
    }
    *outLen = o;
    return i == count;
}

}  // namespace hoststage
}  // namespace kanzi_amd

// C entry points (ctypes in tests/, the stream classes in kanzi_amd.cpp)
extern "C" {
__attribute__((visibility("default"))) int knz_host_text_forward(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* dataType, int* outLen)
{ return kanzi_amd::hoststage::textForward(variant, src, count, dst, dstCap, blockSize, bsVersion, dataType, outLen) ? 1 : 0; }
__attribute__((visibility("default"))) int knz_host_text_inverse(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* outLen)
{ return kanzi_amd::hoststage::textInverse(variant, src, count, dst, dstCap, blockSize, bsVersion, outLen) ? 1 : 0; }
__attribute__((visibility("default"))) int knz_host_utf_forward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* dataType, int* outLen)
{ return kanzi_amd::hoststage::utfForward(src, count, dst, dstCap, dataType, outLen) ? 1 : 0; }
__attribute__((visibility("default"))) int knz_host_utf_inverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* outLen)
{ return kanzi_amd::hoststage::utfInverse(src, count, dst, dstCap, outLen) ? 1 : 0; }
__attribute__((visibility("default"))) int knz_host_preset_data_type(const uint8_t* block, int n)
{ return kanzi_amd::hoststage::presetDataType(block, n); }
}
