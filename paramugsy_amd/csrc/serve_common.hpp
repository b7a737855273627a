// serve_common.hpp -- what the drop-in m_translate (client) and `mugsy_profiles serve -socket` (worker) agree on: where the socket
// lives when nobody says, who may be on the other end of it, and how long either side waits.  Host only, no HIP.
//
// Where: never a bare, predictable name in a world-writable sticky directory (round 4 used /tmp/paramugsy-serve-<uid>.sock: any
// local user could create that path first, the owner's worker could not unlink it, and a client believed whatever answered).  The
// default is $XDG_RUNTIME_DIR/paramugsy/serve.sock when that directory is the caller's, else /tmp/paramugsy-<uid>/serve.sock; the
// directory is created 0700 and must be a real directory (not a link), owned by the caller, closed to group and others -- otherwise
// there is no default and the client runs the job itself.
// Who: both ends read SO_PEERCRED and talk only to their own uid.  An explicit PARAMUGSY_SERVE_SOCKET path is the caller's choice of
// place, but the peer rule holds there too.
#pragma once
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <string>

#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/types.h>
#include <sys/un.h>
#include <unistd.h>

namespace pm_serve {

// a directory that is the caller's alone: exists (or can be made, 0700), is a directory and not a link, st_uid == uid, no group/other bits
inline bool private_dir(const std::string &dir, bool create) {
  struct stat st;
  if(lstat(dir.c_str(), &st) != 0) {
    if(!create || errno != ENOENT || mkdir(dir.c_str(), 0700) != 0) {
      return false;
    }
    if(lstat(dir.c_str(), &st) != 0) {
      return false;
    }
  }
  return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 077) == 0;
}

// "" when no safe default exists
inline std::string default_socket_path(bool create) {
  if(const char *x = getenv("XDG_RUNTIME_DIR")) {
    struct stat st;
    if(*x == '/' && stat(x, &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 077) == 0) {
      const std::string dir = std::string(x) + "/paramugsy";
      if(private_dir(dir, create)) {
        return dir + "/serve.sock";
      }
    }
  }
  const std::string dir = "/tmp/paramugsy-" + std::to_string((long)getuid());
  if(private_dir(dir, create)) {
    return dir + "/serve.sock";
  }
  return std::string();
}

// the uid on the other end of a connected UNIX socket; false when the kernel does not say
inline bool peer_is_me(int fd) {
  struct ucred cred;
  socklen_t len = sizeof cred;
  if(getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cred, &len) != 0 || len != sizeof cred) {
    return false;
  }
  return cred.uid == getuid();
}

inline void set_timeouts(int fd, double recv_seconds, double send_seconds) {
  struct timeval tv;
  tv.tv_sec = (time_t)recv_seconds;
  tv.tv_usec = (suseconds_t)((recv_seconds - (double)tv.tv_sec) * 1e6);
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  tv.tv_sec = (time_t)send_seconds;
  tv.tv_usec = (suseconds_t)((send_seconds - (double)tv.tv_sec) * 1e6);
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}

inline double env_seconds(const char *name, double fallback) {
  const char *e = getenv(name);
  if(!e || !*e) {
    return fallback;
  }
  char *end = nullptr;
  const double v = strtod(e, &end);
  return end != e && v > 0 ? v : fallback;
}

} // namespace pm_serve
