// blades_launch — one process per GPU, torchrun-style rendezvous through the environment.
//
//   blades_launch [--nproc N] [--master-addr A] [--master-port P] [--pin-gpus] [--grace S]
//                 [--stall-timeout S] -- prog args...
//
// Replaces the reference's Ray actor bring-up (fllib/core/execution/worker_group.py:17-62,
// BackendExecutor + placement group) with the plainest thing that works on one 8-GPU node:
// fork N children, give each RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
// (and, with --pin-gpus, HIP_VISIBLE_DEVICES=<local rank>), wait for them, and if one of them
// fails take the rest down (SIGTERM to the exact pids, SIGKILL after the grace period) so a dead
// rank never leaves the others hanging in a collective.  Exit status = first failing child's.
// --stall-timeout S: ranks mark progress in $BLADES_HEARTBEAT_DIR/rank<r> (blades_amd.parallel.runtime.heartbeat,
// once per FL round); a rank that has begun marking and then stays silent for S seconds is taken for hung (stuck
// kernel, deadlocked collective) and the job ends with status 124 — well before the process group's own timeout.
#include <cerrno>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

namespace {

std::vector<pid_t> g_children;
volatile sig_atomic_t g_signal = 0;

void on_signal(int sig) { g_signal = sig; }

void signal_all(int sig) {
  for (pid_t p : g_children)
    if (p > 0) kill(p, sig);
}

int status_to_code(int st) {
  if (WIFEXITED(st)) return WEXITSTATUS(st);
  if (WIFSIGNALED(st)) return 128 + WTERMSIG(st);
  return 1;
}

void usage() {
  fprintf(stderr,
          "usage: blades_launch [--nproc N] [--master-addr ADDR] [--master-port PORT] [--pin-gpus] [--grace SECONDS] "
          "[--stall-timeout SECONDS] -- program [args...]\n");
}

}  // namespace

int main(int argc, char** argv) {
  int nproc = 1, port = 29500, grace = 10, stall = 0;
  bool pin = false;
  std::string addr = "127.0.0.1";
  int i = 1;
  for (; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--") { ++i; break; }
    if (a == "--nproc" && i + 1 < argc) nproc = atoi(argv[++i]);
    else if (a == "--master-addr" && i + 1 < argc) addr = argv[++i];
    else if (a == "--master-port" && i + 1 < argc) port = atoi(argv[++i]);
    else if (a == "--grace" && i + 1 < argc) grace = atoi(argv[++i]);
    else if (a == "--stall-timeout" && i + 1 < argc) stall = atoi(argv[++i]);
    else if (a == "--pin-gpus") pin = true;
    else { usage(); return 2; }
  }
  if (i >= argc || nproc < 1 || nproc > 64) { usage(); return 2; }

  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_signal;
  sigaction(SIGINT, &sa, nullptr);
  sigaction(SIGTERM, &sa, nullptr);

  std::string hb_dir;
  if (stall > 0) {
    char tmpl[] = "/tmp/blades_hb_XXXXXX";
    if (mkdtemp(tmpl) == nullptr) { perror("mkdtemp"); return 1; }
    hb_dir = tmpl;
  }
  g_children.assign(nproc, -1);
  for (int r = 0; r < nproc; ++r) {
    pid_t pid = fork();
    if (pid < 0) {
      perror("fork");
      signal_all(SIGTERM);
      return 1;
    }
    if (pid == 0) {
      const std::string rank = std::to_string(r), world = std::to_string(nproc), p = std::to_string(port);
      setenv("RANK", rank.c_str(), 1);
      setenv("LOCAL_RANK", rank.c_str(), 1);
      setenv("WORLD_SIZE", world.c_str(), 1);
      setenv("LOCAL_WORLD_SIZE", world.c_str(), 1);
      setenv("MASTER_ADDR", addr.c_str(), 1);
      setenv("MASTER_PORT", p.c_str(), 1);
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
      if (!hb_dir.empty()) setenv("BLADES_HEARTBEAT_DIR", hb_dir.c_str(), 1);
      if (pin) {
        setenv("HIP_VISIBLE_DEVICES", rank.c_str(), 1);
        setenv("LOCAL_RANK", "0", 1);
        setenv("BLADES_LOCAL_RANK", rank.c_str(), 1);
      }
      execvp(argv[i], argv + i);
      fprintf(stderr, "blades_launch: cannot exec %s: %s\n", argv[i], strerror(errno));
      _exit(127);
    }
    g_children[r] = pid;
  }

  int exit_code = 0, alive = nproc;
  bool tearing_down = false;
  time_t deadline = 0;
  while (alive > 0) {
    int st = 0;
    pid_t done = waitpid(-1, &st, WNOHANG);
    if (done > 0) {
      for (int r = 0; r < nproc; ++r)
        if (g_children[r] == done) {
          g_children[r] = -1;
          --alive;
          const int code = status_to_code(st);
          if (code != 0) {
            fprintf(stderr, "blades_launch: rank %d (pid %d) ended with status %d\n", r, (int)done, code);
            if (exit_code == 0) exit_code = code;
            if (!tearing_down) {
              tearing_down = true;
              deadline = time(nullptr) + grace;
              signal_all(SIGTERM);
            }
          }
        }
      continue;
    }
    if (done < 0 && errno == ECHILD) break;
    if (g_signal && !tearing_down) {
      tearing_down = true;
      deadline = time(nullptr) + grace;
      if (exit_code == 0) exit_code = 128 + g_signal;
      signal_all(SIGTERM);
    }
    if (stall > 0 && !tearing_down) {
      const time_t now = time(nullptr);
      for (int r = 0; r < nproc; ++r) {
        if (g_children[r] <= 0) continue;
        struct stat sb;
        const std::string path = hb_dir + "/rank" + std::to_string(r);
        if (stat(path.c_str(), &sb) != 0) continue;  // still starting up: the first mark comes from init_runtime
        if (now - sb.st_mtime > stall) {
          fprintf(stderr, "blades_launch: rank %d (pid %d) made no progress for %ld s; ending the job\n", r,
                  (int)g_children[r], (long)(now - sb.st_mtime));
          tearing_down = true;
          deadline = now + grace;
          if (exit_code == 0) exit_code = 124;
          signal_all(SIGTERM);
          break;
        }
      }
    }
    if (tearing_down && time(nullptr) >= deadline) {
      signal_all(SIGKILL);
      deadline = time(nullptr) + 3600;
    }
    struct timespec ts = {0, 50 * 1000 * 1000};
    nanosleep(&ts, nullptr);
  }
  if (!hb_dir.empty()) {
    for (int r = 0; r < nproc; ++r) {
      unlink((hb_dir + "/rank" + std::to_string(r)).c_str());
      unlink((hb_dir + "/rank" + std::to_string(r) + ".tmp").c_str());
    }
    rmdir(hb_dir.c_str());
  }
  return exit_code;
}
