// d2h_write.hip -- where the time of "text to the host and into the file" goes: device-to-host copies of 64 MB in 8 MB pieces into
// pinned memory (hipHostMalloc, default and non-coherent), pwrite of 64 MB from pinned and from ordinary memory with 1-4 threads.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_write tools/ubench/d2h_write.hip -lpthread && /tmp/d2h_write
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// pre: 0 = the file grows as it is written; 1 = ftruncate to the final size first; 2 = posix_fallocate first
static double write_file(const char *src, size_t n, int threads, int pre = 0, const char *path = "/tmp/d2h_write.out") {
  int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
  const double t0 = now();
  if(pre == 1 && ftruncate(fd, (off_t)n) != 0) {
    perror("ftruncate");
  }
  if(pre == 2 && posix_fallocate(fd, 0, (off_t)n) != 0) {
    perror("posix_fallocate");
  }
  std::vector<std::thread> th;
  const size_t piece = 8u << 20;
  for(int w = 0; w < threads; ++w) {
    th.emplace_back([&, w]() {
      for(size_t at = (size_t)w * piece; at < n; at += (size_t)threads * piece) {
        size_t m = std::min(piece, n - at), done = 0;
        while(done < m) {
          ssize_t r = pwrite(fd, src + at + done, m - done, (off_t)(at + done));
          if(r <= 0) {
            return;
          }
          done += (size_t)r;
        }
      }
    });
  }
  for(auto &t : th) {
    t.join();
  }
  const double dt = now() - t0;
  close(fd);
  return dt;
}

int main() {
  const size_t n = 64u << 20, piece = 8u << 20;
  char *dev = nullptr;
  hipMalloc((void **)&dev, n);
  hipMemset(dev, 65, n);
  hipDeviceSynchronize();
  for(int kind = 0; kind < 3; ++kind) {
    char *host = nullptr;
    const char *name = kind == 0 ? "hipHostMalloc default" : (kind == 1 ? "hipHostMalloc non-coherent" : "malloc (pageable)");
    if(kind == 0) {
      hipHostMalloc((void **)&host, n, hipHostMallocPortable);
    }
    else if(kind == 1) {
      hipHostMalloc((void **)&host, n, hipHostMallocPortable | hipHostMallocNonCoherent);
    }
    else {
      host = (char *)malloc(n);
      memset(host, 1, n);
    }
    for(int rep = 0; rep < 3; ++rep) {
      double t0 = now();
      for(size_t at = 0; at < n; at += piece) {
        hipMemcpy(host + at, dev + at, piece, hipMemcpyDeviceToHost);
      }
      double t1 = now();
      hipMemcpy(host, dev, n, hipMemcpyDeviceToHost);
      double t2 = now();
      printf("%-28s D2H 8 x 8 MB blocking %.4f s; one 64 MB copy %.4f s", name, t1 - t0, t2 - t1);
      for(int th = 1; th <= 4; th *= 2) {
        printf("; pwrite %d thr %.4f s", th, write_file(host, n, th));
      }
      printf("\n");
      if(rep == 2) {
        const char *out_dir = getenv("OUT_DIR") ? getenv("OUT_DIR") : "/tmp";
        char path[512];
        snprintf(path, sizeof path, "%s/d2h_write.out", out_dir);
        for(int pre = 0; pre < 3; ++pre) {
          printf("   %s, %s:", path, pre == 0 ? "growing" : (pre == 1 ? "ftruncate first" : "posix_fallocate first"));
          for(int th = 1; th <= 4; th *= 2) {
            printf(" %d thr %.4f s", th, write_file(host, n, th, pre, path));
          }
          printf("\n");
        }
      }
    }
    if(kind < 2) {
      hipHostFree(host);
    }
    else {
      free(host);
    }
  }
  return 0;
}
