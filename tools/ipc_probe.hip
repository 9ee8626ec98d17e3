// Can two PROCESSES sharing one MI355X push into each other's device memory (hipIpcGetMemHandle / hipIpcOpenMemHandle) and order
// the hand-over with interprocess events (hipIpcGetEventHandle / hipIpcOpenEventHandle)?  The capability the peer-push all-gather
// provider (ligero_amd/csrc/push_comm.hip) rests on.  Forks BEFORE any HIP call; the processes talk over two pipes.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ipc_probe tools/ipc_probe.hip && tools/ipc_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[%s] %s: %s\n", who, #call, hipGetErrorString(e_)); return 2; } } while (0)

struct Hello { hipIpcMemHandle_t mem; hipIpcEventHandle_t ev; };

static int rd(int fd, void* p, size_t n) { char* c = (char*)p; while (n) { ssize_t r = read(fd, c, n); if (r <= 0) return -1; c += r; n -= r; } return 0; }
static int wr(int fd, const void* p, size_t n) { const char* c = (const char*)p; while (n) { ssize_t r = write(fd, c, n); if (r <= 0) return -1; c += r; n -= r; } return 0; }

static int run(const char* who, int rank, int fd_in, int fd_out) {
    const size_t N = 1 << 20;                       // u32 words per half; the buffer holds two halves: [rank 0's | rank 1's]
    CK(hipSetDevice(0));
    uint32_t* buf = nullptr;
    CK(hipMalloc(&buf, 2 * N * 4));
    std::vector<uint32_t> h(2 * N, 0);
    for (size_t i = 0; i < N; i++) h[rank * N + i] = 0x10000000u * (rank + 1) + (uint32_t)i;
    CK(hipMemcpy(buf, h.data(), 2 * N * 4, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t pushed;
    CK(hipEventCreateWithFlags(&pushed, hipEventDisableTiming | hipEventInterprocess));
    Hello mine, peer;
    CK(hipIpcGetMemHandle(&mine.mem, buf));
    CK(hipIpcGetEventHandle(&mine.ev, pushed));
    if (wr(fd_out, &mine, sizeof(mine)) || rd(fd_in, &peer, sizeof(peer))) { fprintf(stderr, "[%s] pipe\n", who); return 3; }
    void* peer_buf = nullptr;
    CK(hipIpcOpenMemHandle(&peer_buf, peer.mem, hipIpcMemLazyEnablePeerAccess));
    hipEvent_t peer_pushed;
    CK(hipIpcOpenEventHandle(&peer_pushed, peer.ev));
    // push my half into the peer's buffer, stream-ordered; then the event
    CK(hipMemcpyAsync((uint32_t*)peer_buf + rank * N, buf + rank * N, N * 4, hipMemcpyDeviceToDevice, s));
    CK(hipEventRecord(pushed, s));
    char tok = 1;                                   // host handshake: both have RECORDED before either waits on the other's event
    if (wr(fd_out, &tok, 1) || rd(fd_in, &tok, 1)) return 3;
    CK(hipStreamWaitEvent(s, peer_pushed, 0));
    CK(hipMemcpyAsync(h.data(), buf, 2 * N * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    size_t bad = 0;
    for (int r = 0; r < 2; r++)
        for (size_t i = 0; i < N; i++) bad += h[r * N + i] != 0x10000000u * (r + 1) + (uint32_t)i;
    printf("[%s] all-gather by peer push over HIP IPC: %s (%zu wrong words of %zu)\n", who, bad ? "WRONG" : "ok", bad, 2 * N);
    if (wr(fd_out, &tok, 1) || rd(fd_in, &tok, 1)) return 3;      // nobody unmaps while the other may still read
    CK(hipIpcCloseMemHandle(peer_buf));
    CK(hipEventDestroy(peer_pushed));
    CK(hipEventDestroy(pushed));
    CK(hipFree(buf));
    return bad ? 1 : 0;
}

int main() {
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 9;
    const pid_t pid = fork();                       // before anything touches the GPU
    if (pid == 0) { close(a2b[1]); close(b2a[0]); _exit(run("rank 1", 1, a2b[0], b2a[1])); }
    close(a2b[0]); close(b2a[1]);
    const int rc = run("rank 0", 0, b2a[0], a2b[1]);
    int st = 0;
    waitpid(pid, &st, 0);
    const int rc1 = WIFEXITED(st) ? WEXITSTATUS(st) : 99;
    printf("ipc probe: rank 0 -> %d, rank 1 -> %d\n", rc, rc1);
    return rc | rc1;
}
