// Exercises the C++ host mirror (include/kanzi_amd.hpp) the way the reference's own unit tests
// exercise the reference classes: src/test/TestBWT.cpp, TestTransforms.cpp, TestEntropyCodec.cpp,
// TestCompressedStream.cpp, TestFactories.cpp. Runs on the GPU box (pytest -m gpu drives it) and
// returns 0 / non-zero like the reference's test executables.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "kanzi_amd.hpp"

using namespace kanzi_amd;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static std::vector<byte> gen(int kind, size_t n, unsigned seed)
{
    std::vector<byte> v(n);
    unsigned x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) {
        x = x * 1664525u + 1013904223u;
        switch (kind) {
        case 0: v[i] = byte(x >> 24); break;                                   // random
        case 1: v[i] = byte(65 + ((x >> 24) % 4)); break;                      // small alphabet
        case 2: v[i] = byte((i / 37) & 1 ? 0 : (x >> 28)); break;              // zero heavy
        case 3: v[i] = byte((i * 17 + 3) & 255); break;                        // src/test/test_api.py fill_buffer
        default: v[i] = byte("the quick brown fox jumps over the lazy dog. "[i % 45]); break;
        }
    }
    return v;
}

static void testTransforms()
{
    const char* names[] = { "BWT", "MTFT", "ZRLT", "SRT", "RLT", "LZ", "LZX", "RANK", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT", "RLT+LZX", "BWT+RANK+ZRLT" };
    for (const char* nm : names) {
        for (int kind = 0; kind < 5; kind++) {
            for (size_t n : { size_t(20), size_t(512), size_t(80000) }) {
                std::vector<byte> in = gen(kind, n, unsigned(kind * 7 + n));
                Context ctx;
                ctx.putInt("bsVersion", 6);
                ctx.putString("entropy", "ANS0");
                TransformSequence<byte>* f = TransformFactory<byte>::newTransform(ctx, TransformFactory<byte>::getType(nm));
                std::vector<byte> a(in), b(size_t(f->getMaxEncodedLength(int(n))) + 64), c(n + 2048);
                SliceArray<byte> sa1(a.data(), int(a.size()), 0), sa2(b.data(), int(b.size()), 0), sa3(c.data(), int(c.size()), 0);
                const bool ok = f->forward(sa1, sa2, int(n));
                if (!ok) { CHECK(sa1._index == int(n)); delete f; continue; }   // "does not apply" is accepted (TestTransforms.cpp:1079-1096)
                const int enc = sa2._index;
                TransformSequence<byte>* g = TransformFactory<byte>::newTransform(ctx, TransformFactory<byte>::getType(nm));
                g->setSkipFlags(f->getSkipFlags());
                sa2._index = 0;
                CHECK(g->inverse(sa2, sa3, enc));
                CHECK(sa3._index == int(n) && memcmp(c.data(), in.data(), n) == 0);
                delete f; delete g;
            }
        }
    }
    // BWT known answers (src/test/TestBWT.cpp:42-60)
    {
        const char* s = "mississippi";
        std::vector<byte> in(s, s + 11), out(64);
        BWTBlockCodec bwt;
        SliceArray<byte> sa1(in.data(), 11, 0), sa2(out.data(), 64, 0);
        CHECK(bwt.forward(sa1, sa2, 11));
        CHECK(sa2._index == 13 && out[1] == 4 && memcmp(&out[2], "ipssmpissii", 11) == 0);
    }
    // invalid SliceArray -> std::invalid_argument (src/Transform.hpp contract)
    {
        ZRLT z;
        SliceArray<byte> bad(nullptr, 10, 0);
        std::vector<byte> o(16);
        SliceArray<byte> good(o.data(), 16, 0);
        bool threw = false;
        try { z.forward(bad, good, 4); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
}

static void testEntropy()
{
    const short types[] = { EntropyEncoderFactory::NONE_TYPE, EntropyEncoderFactory::HUFFMAN_TYPE, EntropyEncoderFactory::ANS0_TYPE,
                            EntropyEncoderFactory::ANS1_TYPE, EntropyEncoderFactory::FPAQ_TYPE };
    for (short t : types) {
        for (int kind = 0; kind < 5; kind++) {
            for (size_t n : { size_t(20), size_t(4096), size_t(100000) }) {
                std::vector<byte> in = gen(kind, n, unsigned(kind + n));
                std::stringstream ss;
                Context ctx;
                {
                    DefaultOutputBitStream obs(ss, 16384);
                    obs.writeBits(uint64(5), 3);                       // the codec starts at a non-aligned bit
                    EntropyEncoder* ee = EntropyEncoderFactory::newEncoder(obs, ctx, t);
                    CHECK(ee->encode(in.data(), 0, uint(n)) == int(n));
                    ee->dispose();
                    delete ee;
                    obs.close();
                }
                std::vector<byte> out(n);
                DefaultInputBitStream ibs(ss, 16384);
                CHECK(ibs.readBits(3) == 5);
                EntropyDecoder* ed = EntropyDecoderFactory::newDecoder(ibs, ctx, t);
                CHECK(ed->decode(out.data(), 0, uint(n)) == int(n));
                ed->dispose();
                delete ed;
                CHECK(memcmp(out.data(), in.data(), n) == 0);
            }
        }
    }
}

static void testFactories()
{
    CHECK(TransformFactory<byte>::getType("BWT+MTFT+ZRLT") == 0x47180000000ull);
    CHECK(TransformFactory<byte>::getName(0x47180000000ull) == "BWT+MTFT+ZRLT");
    CHECK(TransformFactory<byte>::getType("none") == 0);
    bool threw = false;
    try { TransformFactory<byte>::getType("A+B"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { TransformFactory<byte>::getType("BWT+BWT+BWT+BWT+BWT+BWT+BWT+BWT+BWT"); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    CHECK(EntropyEncoderFactory::getType("ans0") == 5 && std::string(EntropyEncoderFactory::getName(2)) == "FPAQ");
}

static void testStreams()
{
    // src/test/TestCompressedStream.cpp: sizes 64 KiB .. 4 MiB, several job counts, write/read after close
    struct Cfg { const char* t; const char* e; int bs; } cfgs[] = {
        { "NONE", "ANS0", 65536 }, { "BWT+MTFT+ZRLT", "ANS0", 262144 }, { "RLT+ZRLT", "HUFFMAN", 65536 }, { "BWT+SRT+ZRLT", "FPAQ", 262144 }, { "SRT", "ANS1", 262144 }, { "LZX", "ANS1", 262144 } };
    for (const Cfg& cf : cfgs) {
        for (int jobs = 1; jobs <= 4; jobs += 3) {
            for (size_t n : { size_t(0), size_t(1), size_t(65536), size_t(1000001) }) {
                std::vector<byte> in = gen(int(n % 5), n, unsigned(n + jobs));
                std::stringstream ss;
                if (getenv("KNZ_TEST_VERBOSE")) { printf("stream %s %s bs=%d jobs=%d n=%zu\n", cf.t, cf.e, cf.bs, jobs, n); fflush(stdout); }
                {
                    CompressedOutputStream cos(ss, jobs, cf.e, cf.t, cf.bs);
                    size_t off = 0;
                    while (off < n) { const size_t c = std::min<size_t>(n - off, 77777); cos.write(reinterpret_cast<const char*>(&in[off]), std::streamsize(c)); off += c; }
                    cos.close();
                    bool threw = false;
                    try { cos.write("x", 1); } catch (const IOException& e) { threw = (e.error() == Error::ERR_WRITE_FILE); }
                    CHECK(threw);
                    CHECK(cos.getWritten() == uint64(ss.str().size()));
                }
                std::vector<byte> out(n + 16);
                CompressedInputStream cis(ss, jobs);
                cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
                CHECK(size_t(cis.gcount()) == n);
                CHECK(cis.eof());
                CHECK(n == 0 || memcmp(out.data(), in.data(), n) == 0);
                cis.close();
            }
        }
    }
    // malformed header (src/test/TestMalformedStream.cpp)
    {
        std::string bad(64, '\0');
        std::stringstream ss(bad);
        CompressedInputStream cis(ss, 1);
        char buf[16];
        bool threw = false;
        try { cis.read(buf, 16); } catch (const IOException& e) { threw = (e.error() == Error::ERR_INVALID_FILE); }
        CHECK(threw);
    }
}

// TEXT and UTF run on the host in front of the device chain (levels 5 and 6 of the reference's CLI, app/BlockCompressor.cpp:583-591):
// the two classes by themselves (src/test/TestTransforms.cpp's round trips) and through the stream classes
static void testHostStages()
{
    std::vector<byte> text;
    {
        const char* words[] = { "the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "compression", "transform", "entropy", "block" };
        unsigned x = 99;
        while (text.size() < 300000) {
            x = x * 1664525u + 1013904223u;
            const char* wd = words[(x >> 24) % 12];
            text.insert(text.end(), reinterpret_cast<const byte*>(wd), reinterpret_cast<const byte*>(wd) + strlen(wd));
            text.push_back(byte(((x >> 20) & 15) == 0 ? '\n' : ' '));
        }
    }
    std::vector<byte> utf;
    {
        const char* units[] = { "\xC3\xA9", "\xE2\x82\xAC", "\xD0\x96", "a", "b", " ", "\xF0\x9F\x98\x80", "\xC3\xBC", "\xE4\xB8\xAD" };
        unsigned x = 7;
        while (utf.size() < 120000) { x = x * 1664525u + 1013904223u; const char* u = units[(x >> 24) % 9]; utf.insert(utf.end(), reinterpret_cast<const byte*>(u), reinterpret_cast<const byte*>(u) + strlen(u)); }
    }
    for (const char* ent : { "ANS0", "FPAQ" }) {                              // (the entropy codec picks the TEXT encoding: top-bit or escape-byte indexes)
        Context ctx;
        ctx.putInt("bsVersion", 6);
        ctx.putString("entropy", ent);
        ctx.putInt("blockSize", 1 << 20);
        TextCodec f(ctx), g(ctx);
        std::vector<byte> a(text), b(text.size() + 64), c(text.size() + 64);
        SliceArray<byte> sa1(a.data(), int(a.size()), 0), sa2(b.data(), int(b.size()), 0), sa3(c.data(), int(c.size()), 0);
        CHECK(f.forward(sa1, sa2, int(text.size())));
        const int enc = sa2._index;
        CHECK(enc < int(text.size()));                                        // words become dictionary indexes
        sa2._index = 0;
        CHECK(g.inverse(sa2, sa3, enc));
        CHECK(sa3._index == int(text.size()) && memcmp(c.data(), text.data(), text.size()) == 0);
    }
    {
        Context ctx;
        ctx.putInt("bsVersion", 6);
        UTFCodec f(ctx), g(ctx);
        std::vector<byte> a(utf), b(size_t(f.getMaxEncodedLength(int(utf.size()))) + 64), c(utf.size() + 64);
        SliceArray<byte> sa1(a.data(), int(a.size()), 0), sa2(b.data(), int(b.size()), 0), sa3(c.data(), int(c.size()), 0);
        CHECK(f.forward(sa1, sa2, int(utf.size())));
        const int enc = sa2._index;
        sa2._index = 0;
        CHECK(g.inverse(sa2, sa3, enc));
        CHECK(sa3._index == int(utf.size()) && memcmp(c.data(), utf.data(), utf.size()) == 0);
        // not UTF-8: the codec declines (UTFCodec.cpp: validation of the first bytes)
        std::vector<byte> rnd = gen(0, 50000, 5), o(60000);
        SliceArray<byte> r1(rnd.data(), int(rnd.size()), 0), r2(o.data(), int(o.size()), 0);
        UTFCodec h(ctx);
        CHECK(!h.forward(r1, r2, int(rnd.size())));
    }
    // the two published levels as stream parameters: host stages in front of the device chain, one block per device call
    struct Cfg { const char* t; const char* e; int bs; } cfgs[] = { { "TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 262144 }, { "TEXT+UTF+BWT+SRT+ZRLT", "FPAQ", 262144 }, { "TEXT", "HUFFMAN", 65536 } };
    for (const Cfg& cf : cfgs) {
        for (const std::vector<byte>* in : { &text, &utf }) {
            std::stringstream ss;
            {
                CompressedOutputStream cos(ss, 2, cf.e, cf.t, cf.bs);
                cos.write(reinterpret_cast<const char*>(in->data()), std::streamsize(in->size()));
                cos.close();
            }
            std::vector<byte> out(in->size() + 16);
            CompressedInputStream cis(ss, 2);
            cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
            CHECK(size_t(cis.gcount()) == in->size());
            CHECK(memcmp(out.data(), in->data(), in->size()) == 0);
            cis.close();
        }
    }
}

static void testSeek()
{
    // io/CompressedInputStream.hpp:329-384: block boundaries are the only valid positions; tell() hands them out
    const int bs = 65536;
    const size_t n = 5 * size_t(bs) + 1234;
    std::vector<byte> in = gen(4, n, 7);
    for (int headerless = 0; headerless < 2; headerless++) {
        std::stringstream ss;
        {
            CompressedOutputStream cos(ss, 2, "ANS0", "LZX", bs, 32, 0, headerless != 0);
            cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(n));
            cos.close();
        }
        CompressedInputStream cis(ss, 1, "ANS0", "LZX", bs, 32, 0, headerless != 0);
        cis.setBatchBlocks(1);
        std::vector<int64> bounds;
        std::vector<byte> out(n + 16);
        bounds.push_back(cis.tell());
        CHECK(bounds[0] == (headerless ? 0 : 160));
        size_t off = 0;
        while (off < n) {
            const size_t c = std::min<size_t>(size_t(bs), n - off);
            cis.read(reinterpret_cast<char*>(&out[off]), std::streamsize(c));
            CHECK(size_t(cis.gcount()) == c);
            off += c;
            bounds.push_back(cis.tell());
        }
        CHECK(memcmp(out.data(), in.data(), n) == 0);
        CHECK(bounds.size() == 7);
        const int order[] = { 3, 1, 5, 0, 4 };
        for (int k : order) {
            CHECK(cis.seek(bounds[size_t(k)]));
            const size_t want = std::min<size_t>(size_t(bs), n - size_t(k) * bs);
            std::vector<byte> blk(size_t(bs) + 16);
            cis.read(reinterpret_cast<char*>(blk.data()), std::streamsize(want));
            CHECK(size_t(cis.gcount()) == want);
            CHECK(memcmp(blk.data(), &in[size_t(k) * bs], want) == 0);
            CHECK(cis.tell() == bounds[size_t(k) + 1]);
        }
        // read through the end after a seek, then come back
        CHECK(cis.seek(bounds[4]));
        cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
        CHECK(size_t(cis.gcount()) == n - 4 * size_t(bs) && cis.eof());
        CHECK(cis.seek(bounds[0]));
        cis.read(reinterpret_cast<char*>(out.data()), 100);
        CHECK(cis.gcount() == 100 && memcmp(out.data(), in.data(), 100) == 0);
        CHECK(!cis.seek(-1));
        cis.close();
        CHECK(!cis.seek(bounds[0]) && cis.tell() == -1);
    }
}

static void testBlockRange()
{
    // io/CompressedInputStream.cpp:836-868: with "from" / "to" in the Context only the blocks from <= id < to (1-based) come out
    const int bs = 16384;
    const size_t n = 7 * size_t(bs) + 777;       // 8 blocks
    std::vector<byte> in = gen(4, n, 11);
    std::stringstream ss;
    {
        CompressedOutputStream cos(ss, 1, "HUFFMAN", "BWT+MTFT+ZRLT", bs, 0, 0, false);
        cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(n));
        cos.close();
    }
    const std::string enc = ss.str();
    const int ranges[][2] = { { 1, 0x7FFFFFFF }, { 3, 6 }, { 1, 2 }, { 8, 9 }, { 5, 100 }, { 9, 12 }, { 4, 4 } };
    for (auto& r : ranges) {
        for (int batch = 1; batch <= 3; batch += 2) {
            std::stringstream is(enc);
            Context ctx;
            ctx.putInt("jobs", 2); ctx.putInt("from", r[0]); ctx.putInt("to", r[1]);
            CompressedInputStream cis(is, ctx);
            cis.setBatchBlocks(batch);
            std::vector<byte> out(n + 16);
            cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
            const size_t lo = std::min(n, size_t(r[0] - 1) * bs), hi = std::min(n, size_t(std::max(r[1], r[0]) - 1) * bs);
            CHECK(size_t(cis.gcount()) == hi - lo);
            CHECK(memcmp(out.data(), in.data() + lo, hi - lo) == 0);
        }
    }
}

// The stream classes run threads (worker; reader + decoder) and queue copies: a caller that walks away in the middle, a stream that
// ends in the middle of a batch and a sink that fails must neither hang nor lose their error.
namespace {
class FailingBuf : public std::streambuf {
public:
    explicit FailingBuf(size_t okBytes) : _left(okBytes) {}
protected:
    std::streamsize xsputn(const char*, std::streamsize n) override
    {
        if (size_t(n) > _left) { const std::streamsize k = std::streamsize(_left); _left = 0; return k; }
        _left -= size_t(n);
        return n;
    }
    int_type overflow(int_type ch) override { if (_left == 0) return traits_type::eof(); _left--; return traits_type::not_eof(ch); }
private:
    size_t _left;
};
}

static void testPipeline()
{
    const int bs = 65536;
    const size_t n = 40 * size_t(bs) + 1234;
    std::vector<byte> in = gen(1, n, 99);
    std::stringstream ss;
    {
        CompressedOutputStream cos(ss, 2, "ANS0", "BWT+MTFT+ZRLT", bs);
        cos.setBatchBlocks(4);                                   // 11 batches: every slot and output buffer is reused several times
        size_t off = 0;
        while (off < n) { const size_t c = std::min<size_t>(n - off, 100000); cos.write(reinterpret_cast<const char*>(&in[off]), std::streamsize(c)); off += c; }
        cos.close();
        CHECK(cos.getWritten() == uint64(ss.str().size()));
    }
    const std::string knz = ss.str();
    // (1) complete read, small batches
    {
        std::stringstream is(knz);
        CompressedInputStream cis(is, 2);
        cis.setBatchBlocks(3);
        std::vector<byte> out(n + 16);
        cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
        CHECK(size_t(cis.gcount()) == n && memcmp(out.data(), in.data(), n) == 0);
    }
    // (2) the caller reads a little and walks away while the threads are ahead of it
    for (int rep = 0; rep < 3; rep++) {
        std::stringstream is(knz);
        CompressedInputStream cis(is, 2);
        cis.setBatchBlocks(2);
        std::vector<byte> out(100000);
        cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size() >> rep));
        CHECK(memcmp(out.data(), in.data(), out.size() >> rep) == 0);
        if (rep == 1) cis.close();
    }
    // (3) the stream ends in the middle of a later batch: everything before it is delivered, then the error
    {
        std::stringstream is(knz.substr(0, knz.size() * 2 / 3));
        CompressedInputStream cis(is, 2);
        cis.setBatchBlocks(3);
        std::vector<byte> out(n + 16);
        bool threw = false;
        size_t got = 0;
        try {
            for (;;) { cis.read(reinterpret_cast<char*>(out.data()) + got, 50000); const size_t g = size_t(cis.gcount()); got += g; if (g == 0) break; }
        } catch (const IOException&) { threw = true; }
        CHECK(threw);
        CHECK(got >= size_t(bs) && got < n && memcmp(out.data(), in.data(), got) == 0);
    }
    // (4) a sink that stops taking bytes: the writer reports it (from write() or close()), and the object can be destroyed
    {
        FailingBuf fb(knz.size() / 2);
        std::ostream os(&fb);
        bool threw = false;
        try {
            CompressedOutputStream cos(os, 2, "ANS0", "BWT+MTFT+ZRLT", bs);
            cos.setBatchBlocks(4);
            size_t off = 0;
            while (off < n) { const size_t c = std::min<size_t>(n - off, 100000); cos.write(reinterpret_cast<const char*>(&in[off]), std::streamsize(c)); off += c; }
            cos.close();
        } catch (const IOException& e) { threw = (e.error() == Error::ERR_WRITE_FILE); }
        CHECK(threw);
    }
    // (5) a writer that is dropped without close()
    {
        std::stringstream s2;
        { CompressedOutputStream cos(s2, 2, "ANS0", "NONE", bs); cos.setBatchBlocks(2); cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(5 * bs + 7)); }
        std::stringstream is(s2.str());
        CompressedInputStream cis(is, 1);
        std::vector<byte> out(5 * bs + 64);
        cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
        CHECK(size_t(cis.gcount()) == size_t(5 * bs + 7) && memcmp(out.data(), in.data(), size_t(5 * bs + 7)) == 0);
    }
}

// The call forms the reference's own callers use (boundary row): src/api/Compressor.cpp:230-237 and
// src/api/Decompressor.cpp:159-166 (positional nullptr for the thread pool), src/app/BlockCompressor.cpp:757 /
// BlockDecompressor.cpp (Context form + addListener), io/CompressedOutputStream.hpp:228-243, CompressedInputStream.hpp:306-327,
// and the codec constructors with their trailing parameters (entropy/HuffmanDecoder.hpp:32, HuffmanEncoder.hpp:32,
// ANSRangeEncoder.hpp:48-51, ANSRangeDecoder.hpp:45-47).
struct CountingListener : public Listener<Event> {
    int seen = 0;
    void processEvent(const Event&) { seen++; }
};

static void testReferenceCallForms()
{
    const int bs = 65536;
    std::vector<byte> in = gen(4, size_t(3 * bs + 1234), 77);
    std::string viaPool, viaCtx, plain;
    {   // Compressor.cpp:230-237
        std::stringstream ss;
        OutputStream& fos = ss;
        CompressedOutputStream* pCos = new CompressedOutputStream(fos, 2, "ANS0", "BWT+MTFT+ZRLT", bs, 32, uint64(in.size()),
                                                                  nullptr,
                                                                  false);
        pCos->write(reinterpret_cast<const char*>(in.data()), std::streamsize(in.size()));
        CHECK(&pCos->flush() == pCos);                                   // NOOP, returns the stream
        bool threw = false;
        try { pCos->tellp(); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { pCos->seekp(0); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw);
        pCos->close();
        delete pCos;
        viaPool = ss.str();
    }
    {   // BlockCompressor.cpp:757: Context form, listeners added right after construction
        std::stringstream ss;
        Context ctx;
        ctx.putInt("jobs", 2); ctx.putInt("blockSize", bs); ctx.putString("entropy", "ANS0"); ctx.putString("transform", "BWT+MTFT+ZRLT");
        ctx.putInt("checksum", 32); ctx.putLong("fileSize", int64(in.size()));
        CountingListener l1, l2;
        CompressedOutputStream cos(ss, ctx);
        CHECK(cos.addListener(l1));
        CHECK(cos.addListener(l2));
        CHECK(cos.removeListener(l1));
        CHECK(!cos.removeListener(l1));                                  // CompressedOutputStream.cpp:350-359
        cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(in.size()));
        cos.close();
        viaCtx = ss.str();
        bool threw = false;
        Context bad;                                                      // no block size: the reference's message
        try { CompressedOutputStream c2(ss, bad); } catch (const std::invalid_argument& e) { threw = std::string(e.what()).find("block size must be at least") != std::string::npos; }
        CHECK(threw);
    }
    {
        std::stringstream ss;
        CompressedOutputStream cos(ss, 2, "ANS0", "BWT+MTFT+ZRLT", bs, 32, uint64(in.size()));
        cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(in.size()));
        cos.close();
        plain = ss.str();
    }
    CHECK(!plain.empty() && viaPool == plain && viaCtx == plain);
    {   // Decompressor.cpp:159-166: headerless form with the pool positional and the bitstream version behind it
        std::stringstream hs;
        {
            CompressedOutputStream cos(hs, 1, "HUFFMAN", "RLT", bs, 0, 0, nullptr, true);
            cos.write(reinterpret_cast<const char*>(in.data()), std::streamsize(in.size()));
            cos.close();
        }
        InputStream& fis = hs;
        CompressedInputStream* pCis = new CompressedInputStream(fis, 1, "HUFFMAN", "RLT", bs, 0, uint64(in.size()),
                                                                nullptr,
                                                                true, 6);
        std::vector<byte> out(in.size() + 8);
        pCis->read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
        CHECK(size_t(pCis->gcount()) == in.size() && memcmp(out.data(), in.data(), in.size()) == 0);
        bool threw = false;
        try { pCis->tellg(); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { pCis->seekg(0); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { pCis->putback('x'); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw && pCis->bad());
        pCis->clear();
        threw = false;
        try { pCis->unget(); } catch (const std::ios_base::failure&) { threw = true; }
        CHECK(threw && pCis->bad());
        pCis->close();
        delete pCis;
    }
    {   // Context form of the reader + listeners, default stream (header present)
        std::stringstream ss(plain);
        Context ctx;
        ctx.putInt("jobs", 3);
        CountingListener l;
        CompressedInputStream cis(ss, ctx);
        CHECK(cis.addListener(l) && cis.removeListener(l) && !cis.removeListener(l));
        std::vector<byte> out(in.size() + 8);
        cis.read(reinterpret_cast<char*>(out.data()), std::streamsize(out.size()));
        CHECK(size_t(cis.gcount()) == in.size() && memcmp(out.data(), in.data(), in.size()) == 0);
    }
    // codec constructors: trailing parameters as the reference declares them; the reference's range checks and messages
    {
        std::stringstream ss;
        std::vector<byte> blk = gen(1, 50000, 5);
        {
            DefaultOutputBitStream obs(ss);
            HuffmanEncoder he(obs, HuffmanCommon::MAX_CHUNK_SIZE);
            CHECK(he.encode(blk.data(), 0, uint(blk.size())) == int(blk.size()));
            ANSRangeEncoder ae(obs, 0, 16384, 12);
            CHECK(ae.encode(blk.data(), 0, uint(blk.size())) == int(blk.size()));
            ANSRangeEncoder a1(obs, 1, 16384, 12);
            CHECK(a1.encode(blk.data(), 0, uint(blk.size())) == int(blk.size()));
            obs.close();
        }
        DefaultInputBitStream ibs(ss);
        Context ctx;
        ctx.putInt("bsVersion", 6);
        std::vector<byte> out(blk.size());
        HuffmanDecoder hd(ibs, &ctx, HuffmanCommon::MAX_CHUNK_SIZE);
        CHECK(hd.decode(out.data(), 0, uint(out.size())) == int(out.size()) && out == blk);
        std::fill(out.begin(), out.end(), byte(0));
        ANSRangeDecoder ad(ibs, 0, 16384);
        CHECK(ad.decode(out.data(), 0, uint(out.size())) == int(out.size()) && out == blk);
        std::fill(out.begin(), out.end(), byte(0));
        ANSRangeDecoder ad1(ibs, 1, 16384);
        CHECK(ad1.decode(out.data(), 0, uint(out.size())) == int(out.size()) && out == blk);
        auto throwsWith = [](auto&& f, const char* what) {
            try { f(); } catch (const std::invalid_argument& e) { return std::string(e.what()).find(what) != std::string::npos; }
            return false;
        };
        std::stringstream s2;
        DefaultOutputBitStream o2(s2);
        DefaultInputBitStream i2(s2);
        CHECK(throwsWith([&] { HuffmanEncoder x(o2, 512); }, "at least 1024"));
        CHECK(throwsWith([&] { HuffmanEncoder x(o2, 1 << 15); }, "at most 16384"));
        CHECK(throwsWith([&] { HuffmanDecoder x(i2, nullptr, 512); }, "at least 1024"));
        CHECK(throwsWith([&] { ANSRangeEncoder x(o2, 2); }, "order must be 0 or 1"));
        CHECK(throwsWith([&] { ANSRangeEncoder x(o2, 0, 512); }, "at least 1024"));
        CHECK(throwsWith([&] { ANSRangeEncoder x(o2, 0, (1 << 27) + 1); }, "at most"));
        CHECK(throwsWith([&] { ANSRangeEncoder x(o2, 0, 16384, 16); }, "Invalid range: 16"));
        CHECK(throwsWith([&] { ANSRangeDecoder x(i2, 0, 100); }, "at least 1024"));
        CHECK(throwsWith([&] { ANSRangeEncoder x(o2, 0, 32768); }, "default chunk size"));      // valid for the reference, no kernel here: refused loudly
        o2.close();
    }
}

// A sink that throws (a streambuf whose medium is full, an ostream with exceptions() set): the runs are appended by the sink thread
// (kanzi_amd.hpp, CompressedOutputStream), and what the sink throws there must reach the caller's thread as an exception of
// write() / close() -- not leave the thread function (std::terminate). The same with the sink on the caller's thread.
struct FullBuf : std::streambuf {
    size_t room;
    explicit FullBuf(size_t n) : room(n) {}
    std::streamsize xsputn(const char*, std::streamsize n) override
    {
        if (size_t(n) > room) throw std::runtime_error("medium full");
        room -= size_t(n);
        return n;
    }
    int overflow(int c) override { if (room == 0) throw std::runtime_error("medium full"); room--; return c; }
};

static void testThrowingSink()
{
    const std::vector<byte> data = gen(0, 40 * 16384, 5);              // incompressible: every batch is larger than the sink's room
    for (int onThread = 0; onThread < 2; onThread++) {
        setenv("KNZ_SINK_THREAD", onThread ? "1" : "0", 1);
        setenv("KNZ_BATCH_BLOCKS", "2", 1);
        for (int mode = 0; mode < 2; mode++) {                        // 0: the streambuf throws, 1: failbit + exceptions()
            FullBuf fb(20000);
            std::ostream os(&fb);
            if (mode) os.exceptions(std::ios::failbit | std::ios::badbit);
            bool caught = false;
            try {
                CompressedOutputStream cos(os, 1, "NONE", "NONE", 16384, 0, 0);
                for (size_t off = 0; off < data.size(); off += 16384) cos.write(reinterpret_cast<const char*>(&data[off]), 16384);
                cos.close();
            } catch (const std::exception&) {
                caught = true;
            }
            CHECK(caught);
        }
    }
    unsetenv("KNZ_SINK_THREAD");
    unsetenv("KNZ_BATCH_BLOCKS");
    // the pools hold what these streams left behind; a second call finds nothing
    CHECK(releaseIdleBuffers() > 0);
    CHECK(releaseIdleBuffers() == 0);
}

int main(int argc, char** argv)
{
    const std::string what = argc > 1 ? argv[1] : "all";
    try {
        if (what == "all" || what == "factories") testFactories();
        if (what == "all" || what == "transforms") testTransforms();
        if (what == "all" || what == "entropy") testEntropy();
        if (what == "all" || what == "streams") testStreams();
        if (what == "all" || what == "callforms") testReferenceCallForms();
        if (what == "all" || what == "hoststages") testHostStages();
        if (what == "all" || what == "seek") testSeek();
        if (what == "all" || what == "range") testBlockRange();
        if (what == "all" || what == "pipeline") testPipeline();
        if (what == "all" || what == "sink") testThrowingSink();
    } catch (const std::exception& e) {
        printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
    return fails ? 1 : 0;
}
