// read_modes.cpp -- what does it cost to bring 10 000 TSV files (1.8 GB, page cache) into one arena?
//   0: read() straight into the arena (what the streamed scan does)      1: read() into a small per-thread buffer, reused
//   2: as 1, then a copy into the arena with non-temporal stores          3: as 1, then memcpy into the arena
//   g++ -O3 -std=c++17 -mavx2 scripts/micro/read_modes.cpp -lpthread -o scripts/micro/read_modes
//   read_modes <dir> <threads>
#include <immintrin.h>
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>
static double proc_cpu() { timespec ts{}; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void nt_copy(char *dst, const char *src, size_t n)
{
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31)) { dst[i] = src[i]; ++i; }
    for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i *)(dst + i), _mm256_loadu_si256((const __m256i *)(src + i)));
    for (; i < n; ++i) dst[i] = src[i];
    _mm_sfence();
}
int main(int argc, char **argv)
{
    const char *dir = argv[1];
    const int nt = atoi(argv[2]);
    std::vector<std::string> files;
    DIR *d = opendir(dir);
    while (dirent *e = readdir(d)) if (strstr(e->d_name, ".tsv")) files.push_back(std::string(dir) + "/" + e->d_name);
    closedir(d);
    std::sort(files.begin(), files.end());
    size_t total = 0;
    std::vector<size_t> off(files.size() + 1, 0);
    for (size_t i = 0; i < files.size(); ++i) { struct stat sb; stat(files[i].c_str(), &sb); off[i + 1] = off[i] + (size_t)sb.st_size + 64; }
    total = off.back();
    const size_t len = (total + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    char *arena = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(arena, len, MADV_HUGEPAGE);
    memset(arena, 1, len);
    for (int mode : {0, 1, 2, 3, 0, 1, 2, 3}) {
        std::atomic<size_t> next{0};
        const double c0 = proc_cpu(), t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&] {
            std::vector<char> buf(1 << 20);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= files.size()) break;
                const size_t n = off[i + 1] - off[i] - 64;
                const int fd = open(files[i].c_str(), O_RDONLY | O_CLOEXEC);
                char *dst = mode == 0 ? arena + off[i] : buf.data();
                if (mode != 0 && n > buf.size()) buf.resize(n), dst = buf.data();
                size_t got = 0;
                while (got < n) { ssize_t r = read(fd, dst + got, n - got); if (r <= 0) break; got += (size_t)r; }
                close(fd);
                if (mode == 2) nt_copy(arena + off[i], buf.data(), n);
                if (mode == 3) memcpy(arena + off[i], buf.data(), n);
            }
        });
        for (auto &t : th) t.join();
        const double w = now() - t0, c = proc_cpu() - c0;
        std::printf("mode %d: %zu files, %.2f GB, %d threads: wall %.1f ms, CPU %.3f s (%.1f us per file, %.2f GB/s per CPU)\n", mode, files.size(),
                    total / 1e9, nt, w * 1e3, c, c / files.size() * 1e6, total / 1e9 / c);
    }
}
