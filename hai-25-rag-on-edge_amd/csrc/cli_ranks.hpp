// cli_ranks.hpp -- "one process per GPU" for the C++ CLIs.
//
// `--gpus N` (N > 1) makes the CLI fork N - 1 children BEFORE anything initialises HIP; the parent is rank 0 and the
// only rank that writes result files.  Rank r uses device r.  The RCCL bootstrap id travels from rank 0 to the children
// through pipes created before the fork; every rank then builds its vs_comm (collective).  The reference is single
// device (one phone); this is the harness side of SURVEY.md 8(e).
#pragma once
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vsearch.h"

namespace vsearch {

struct RankSet {
    int rank = 0, world = 1;
    vs_comm* comm = nullptr;
    std::vector<pid_t> children;  // rank 0 only
    std::vector<int> id_wr;       // rank 0: write ends of the id pipes
    int id_rd = -1;               // rank > 0: read end
};

// integer command-line argument; a bad one is a usage error, not an uncaught std::invalid_argument
inline bool arg_int(const char* text, int& out) {
    char* end = nullptr;
    errno = 0;
    const long v = std::strtol(text, &end, 10);
    if (errno != 0 || end == text || *end != '\0' || v < -2147483647L || v > 2147483647L) return false;
    out = (int)v;
    return true;
}

// removes "--gpus N" from argv (anywhere) and returns N (1 when absent)
inline int take_gpus_flag(int& argc, char** argv) {
    int n = 1;
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "--gpus" && i + 1 < argc) {
            if (!arg_int(argv[i + 1], n)) throw std::runtime_error("--gpus needs an integer");
            for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2];
            argc -= 2;
            break;
        }
    if (n < 1 || n > 64) throw std::runtime_error("--gpus must be in 1..64");
    return n;
}

// Failure propagation.  Once the ranks are forked a rank that dies (device missing, file error, uncaught exception) would
// leave the others blocked in ncclCommInitRank / ncclAllGather for ever.  Rank 0 therefore watches its children from a
// thread: a child that exits abnormally takes the job down (the other children are killed, rank 0 exits non-zero); the
// children ask the kernel for SIGKILL when rank 0 disappears.
struct ChildWatch {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
};
inline ChildWatch& child_watch() {
    static ChildWatch w;
    return w;
}
inline void watch_children(const std::vector<pid_t>& kids) {
    if (kids.empty()) {
        child_watch().done = true;
        return;
    }
    std::thread([kids]() {
        size_t left = kids.size();
        while (left > 0) {
            int st = 0;
            const pid_t p = waitpid(-1, &st, 0);
            if (p < 0) {
                if (errno == EINTR) continue;
                break;
            }
            if (std::find(kids.begin(), kids.end(), p) == kids.end()) continue;
            --left;
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
                std::fprintf(stderr, "a rank process (pid %d) failed: stopping the job\n", (int)p);
                for (pid_t k : kids)
                    if (k != p) kill(k, SIGKILL);
                _exit(1);
            }
        }
        ChildWatch& w = child_watch();
        std::lock_guard<std::mutex> lk(w.mu);
        w.done = true;
        w.cv.notify_all();
    }).detach();
}

// must run before ANY HIP call (vs_device_count included): a forked child cannot inherit an initialised runtime
inline RankSet fork_ranks(int world) {
    RankSet rs;
    rs.world = world;
    for (int r = 1; r < world; ++r) {
        int fd[2];
        if (pipe(fd) != 0) throw std::runtime_error("pipe() failed");
        const pid_t pid = fork();
        if (pid < 0) throw std::runtime_error("fork() failed");
        if (pid == 0) {  // child = rank r: keeps the read end, says nothing on stdout
            close(fd[1]);
            for (int w : rs.id_wr) close(w);
            rs.id_wr.clear();
            rs.children.clear();
            rs.rank = r;
            rs.id_rd = fd[0];
            prctl(PR_SET_PDEATHSIG, SIGKILL);  // rank 0 gone (crash, kill): do not wait in a collective for ever
            const int devnull = open("/dev/null", O_WRONLY);
            if (devnull >= 0) {
                dup2(devnull, STDOUT_FILENO);
                close(devnull);
            }
            return rs;
        }
        close(fd[0]);
        rs.children.push_back(pid);
        rs.id_wr.push_back(fd[1]);
    }
    watch_children(rs.children);  // (rank 0, after the last fork)
    return rs;
}

// collective: rank 0 generates the RCCL id and ships it; every rank creates its communicator on device `rank`
inline void connect_ranks(RankSet& rs) {
    if (rs.world == 1) return;
    char id[VS_COMM_ID_BYTES];
    if (rs.rank == 0) {
        if (vs_comm_unique_id(id) != VS_OK) throw std::runtime_error(vs_last_error());
        for (int w : rs.id_wr) {
            if (write(w, id, sizeof(id)) != (ssize_t)sizeof(id)) throw std::runtime_error("cannot send the RCCL id to a rank");
            close(w);
        }
        rs.id_wr.clear();
    } else {
        size_t got = 0;
        while (got < sizeof(id)) {
            const ssize_t n = read(rs.id_rd, id + got, sizeof(id) - got);
            if (n <= 0) throw std::runtime_error("rank 0 went away before sending the RCCL id");
            got += (size_t)n;
        }
        close(rs.id_rd);
        rs.id_rd = -1;
    }
    if (vs_comm_create(id, rs.rank, rs.world, rs.rank, &rs.comm) != VS_OK) throw std::runtime_error(vs_last_error());
}

// rank 0 waits for the children; returns non-zero if any rank failed
inline int join_ranks(RankSet& rs, int my_status) {
    if (rs.comm) {
        vs_comm_destroy(rs.comm);
        rs.comm = nullptr;
    }
    int status = my_status;
    for (int w : rs.id_wr) close(w);  // a rank 0 that failed early must not leave the children blocked on the pipe
    rs.id_wr.clear();
    if (!rs.children.empty()) {
        if (status != 0) {  // rank 0 failed: the children may sit in a collective that will never complete
            for (pid_t pid : rs.children) kill(pid, SIGKILL);
        }
        // (the watcher thread reaps the children; a child that failed has taken the process down with exit code 1 already)
        ChildWatch& w = child_watch();
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
    }
    rs.children.clear();
    return status;
}

// contiguous row shards for brute force; interior bounds are multiples of 16 (one MFMA tile)
inline int64_t shard_bound(int64_t n_rows, int world, int i) {
    if (i <= 0) return 0;
    if (i >= world) return n_rows;
    return (n_rows * i / world) / 16 * 16;
}

}  // namespace vsearch
