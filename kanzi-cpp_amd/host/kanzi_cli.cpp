// kanzi_amd_cli -- a command line front end over the MI355X block pipeline with the reference CLI's option names
// (src/app/Kanzi.cpp:404-965; level table src/app/BlockCompressor.cpp:556-613). One file in, one file out.
//
//   kanzi_amd_cli -c -i FILE [-o FILE.knz] [-t TRANSFORMS] [-e ENTROPY] [-l LEVEL] [-b SIZE] [-j JOBS] [-x | -x32 | -x64] [-f]
//   kanzi_amd_cli -d -i FILE.knz [-o FILE] [-j JOBS] [--from=N] [--to=N] [-f]
//
// Levels 0, 1, 5 and 6 are in (5 and 6: TEXT and UTF on the host in front of the device chain, host/text_codec.cpp).
// What it does not do (and says so instead of guessing): directories, stdin/stdout, `-y` info, levels whose chains need the
// reference's CPU-only transforms (EXE, PACK, MM, DNA, ROLZ, LZP) or entropy coders (CM, TPAQ): levels 2-4 and 7-9.
// Files written here are byte-identical to `kanzi -c` with the same -t/-e/-b/-x/-j, and either tool reads the other's files
// (tests/test_host_stub.py, tests/test_gpu_host_api.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "kanzi_amd.hpp"

using namespace kanzi_amd;

static bool startsWith(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }

static long long parseSize(const std::string& v)
{
    if (v.empty()) return -1;
    char* end = nullptr;
    long long n = strtoll(v.c_str(), &end, 10);
    if (end == v.c_str()) return -1;
    const std::string suf(end);
    if (suf == "k" || suf == "K") n <<= 10; else if (suf == "m" || suf == "M") n <<= 20; else if (suf == "g" || suf == "G") n <<= 30; else if (!suf.empty()) return -1;
    return n;
}

static int usage(const char* msg)
{
    if (msg) fprintf(stderr, "%s\n", msg);
    fprintf(stderr, "usage: kanzi_amd_cli -c|-d -i FILE [-o FILE] [-t TRANSFORMS] [-e ENTROPY] [-l 0|1|5|6] [-b SIZE] [-j JOBS] [-x|-x32|-x64] [--from=N] [--to=N] [-f]\n");
    return Error::ERR_MISSING_PARAM;
}

int main(int argc, char** argv)
{
    std::string mode, in, out, transform, entropy;
    long long block = 4 << 20;
    bool blockGiven = false;
    int jobs = 1, checksum = 0, level = -1, from = 1, to = 0x7FFFFFFF;
    bool force = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](const char* shortOpt, const char* longOpt, std::string& dst) -> bool {
            if (a == shortOpt) { if (i + 1 >= argc) return false; dst = argv[++i]; return true; }
            if (startsWith(a, longOpt)) { dst = a.substr(strlen(longOpt)); return true; }
            return false;
        };
        std::string v;
        if (a == "-c" || a == "--compress") mode = "c";
        else if (a == "-d" || a == "--decompress") mode = "d";
        else if (a == "-f" || a == "--force") force = true;
        else if (a == "-x" || a == "-x32") checksum = 32;
        else if (a == "-x64") checksum = 64;
        else if (startsWith(a, "--checksum=")) checksum = atoi(a.c_str() + 11);
        else if (val("-i", "--input=", in)) {}
        else if (val("-o", "--output=", out)) {}
        else if (val("-t", "--transform=", transform)) {}
        else if (val("-e", "--entropy=", entropy)) {}
        else if (val("-b", "--block=", v)) { block = parseSize(v); blockGiven = true; if (block < 0) return usage("invalid block size"); }
        else if (val("-j", "--jobs=", v)) jobs = atoi(v.c_str());
        else if (val("-l", "--level=", v)) level = atoi(v.c_str());
        else if (val("-v", "--verbose=", v)) {}
        else if (startsWith(a, "--from=")) from = atoi(a.c_str() + 7);
        else if (startsWith(a, "--to=")) to = atoi(a.c_str() + 5);
        else return usage(("unknown option " + a).c_str());
    }
    if (mode.empty() || in.empty()) return usage(nullptr);
    if (level >= 0) {
        // BlockCompressor.cpp:556-613
        if (level == 0) { transform = "NONE"; entropy = "NONE"; }
        else if (level == 1) { transform = "LZX"; entropy = "NONE"; }
        else if (level == 5) { transform = "TEXT+UTF+BWT+RANK+ZRLT"; entropy = "ANS0"; }      // TEXT and UTF run on the host, the rest on the device
        else if (level == 6) { transform = "TEXT+UTF+BWT+SRT+ZRLT"; entropy = "FPAQ"; if (!blockGiven) block = 8 << 20; }      // (BlockCompressor.cpp:121-124: 8 MiB blocks by default)
        else { fprintf(stderr, "level %d needs transforms or entropy coders that only exist in the CPU reference (EXE/PACK/MM/DNA/ROLZ/LZP, CM/TPAQ); use -t/-e\n", level); return Error::ERR_INVALID_CODEC; }
    }
    if (transform.empty()) transform = "NONE";
    if (entropy.empty()) entropy = "NONE";
    if (out.empty()) {
        if (mode == "c") out = in + ".knz";
        else out = (in.size() > 4 && in.compare(in.size() - 4, 4, ".knz") == 0) ? in.substr(0, in.size() - 4) : in + ".bak";
    }
    struct stat st;
    if (stat(in.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { fprintf(stderr, "cannot read %s (a regular file is required)\n", in.c_str()); return Error::ERR_OPEN_FILE; }
    struct stat so;
    if (!force && stat(out.c_str(), &so) == 0) { fprintf(stderr, "%s exists (use -f)\n", out.c_str()); return Error::ERR_OVERWRITE_FILE; }
    std::ifstream is(in, std::ios::binary);
    std::ofstream os(out, std::ios::binary | std::ios::trunc);
    if (!is || !os) { fprintf(stderr, "cannot open the files\n"); return Error::ERR_OPEN_FILE; }
    std::vector<char> buf(size_t(8) << 20);
    try {
        if (mode == "c") {
            // the reference CLI rounds the block size up to 16 and stores the input size in the header (BlockCompressor.cpp)
            const int bs = int((block + 15) & ~15ll);
            CompressedOutputStream cos(os, jobs, entropy, transform, bs, checksum, uint64(st.st_size), false);
            for (;;) {
                is.read(buf.data(), std::streamsize(buf.size()));
                const std::streamsize got = is.gcount();
                if (got <= 0) break;
                cos.write(buf.data(), got);
            }
            cos.close();
            os.flush();
            fprintf(stderr, "%lld -> %llu bytes\n", (long long)st.st_size, (unsigned long long)cos.getWritten());
        } else {
            Context ctx;
            ctx.putInt("jobs", jobs); ctx.putInt("from", from); ctx.putInt("to", to);
            CompressedInputStream cis(is, ctx);
            unsigned long long total = 0;
            for (;;) {
                cis.read(buf.data(), std::streamsize(buf.size()));
                const std::streamsize got = cis.gcount();
                if (got <= 0) break;
                os.write(buf.data(), got);
                total += (unsigned long long)got;
            }
            cis.close();
            os.flush();
            fprintf(stderr, "%lld -> %llu bytes\n", (long long)st.st_size, total);
        }
    } catch (const IOException& e) {
        fprintf(stderr, "error %d: %s\n", e.error(), e.what());
        return e.error();
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return Error::ERR_UNKNOWN;
    }
    return 0;
}
