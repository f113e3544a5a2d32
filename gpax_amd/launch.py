"""
One process per GPU without PyTorch: the rendezvous and launcher plumbing around `_lib.Rank` (include/gpx.h gpx_rank_*).

A launcher (`python -m torch.distributed.run`, `mpirun`, or `spawn_ranks` below) starts N processes with RANK,
LOCAL_RANK and WORLD_SIZE in their environment.  What the processes still have to agree on before they can form an RCCL
communicator is rank 0's 128-byte unique id (ncclGetUniqueId -> ncclCommInitRank); on one node that travels through a
small directory of files all ranks can see (`FileStore`: write to a temporary name, rename = atomic publish, readers
poll).  The same store carries the agreement on the transport: if RCCL cannot be initialised on any rank, every rank
falls back to the library's "file" transport together — and if an initialisation HANGS (a peer died inside
ncclCommInitRank), a watchdog re-executes the process with the fallback selected, so a run never sits in a dead
collective.

Reference seam: device placement / chain_method of gpax (gpax/models/gp.py:173-174,201-203) and the vmap over posterior
samples (gp.py:392-395) that the ranks shard.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class RankEnv:
    rank: int
    world: int
    local_rank: int
    rdzv_dir: str
    attempt: int = 0
    owns_dir: bool = False  # the directory was derived here (not handed over by a launcher that cleans it up)
    token: str = ""         # names THIS launch inside the directory (see launch_token)


def launch_token(environ=None) -> str:
    """A name every rank of ONE launch derives alike and that no other launch shares, so that a rendezvous directory
    that is used again (a GPX_RDZV_DIR set by hand, a crashed run, a launcher restarting its workers under the same
    port) never shows this launch the keys or transfer files of an earlier one: $GPX_RDZV_TOKEN when the launcher hands
    one out (spawn_ranks does), else the launcher's agent process — the common parent of the ranks — by PID AND start
    time, plus what the launcher exports about the run."""
    e = os.environ if environ is None else environ
    t = e.get("GPX_RDZV_TOKEN")
    if t:
        return "".join(c if (c.isalnum() or c in "-_.") else "_" for c in t)[:96]
    ppid, start = os.getppid(), "0"
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time of the process, in clock ticks since boot
    except (OSError, IndexError):
        pass
    run = "".join(c if c.isalnum() else "_" for c in e.get("TORCHELASTIC_RUN_ID", ""))[:32]
    # an elastic agent that restarts its workers keeps its PID, port and run id: the restart count tells the
    # incarnations apart (a restarted rank must not read the dead incarnation's unique id and wait for its peers)
    restart = "".join(c for c in e.get("TORCHELASTIC_RESTART_COUNT", "0") if c.isdigit())[:8] or "0"
    return f"p{ppid}_{start}_{e.get('MASTER_PORT', '0')}_{run}_r{restart}"


def _private_dir(path: str, parent: bool = False) -> None:
    """Create `path` for this user only and refuse one that somebody else could have planted (another owner, a symbolic
    link, write access for group / others): the directory carries the 128-byte RCCL bootstrap id.
    parent = True: `path` is the rendezvous directory itself, possibly handed over by the user (GPX_RDZV_DIR made by hand
    under umask 002 is 0775): owner, not-a-symlink and — unless the sticky bit is set, as on /tmp — no write access for
    OTHERS are checked there (whoever can write to it could rename the <token> subdirectory between the checks and plant
    their own); the ids live in the <token>/attempt subdirectories this module creates itself, and those are held to the
    full write-bit rule."""
    os.makedirs(path, mode=0o700, exist_ok=True)
    st = os.lstat(path)
    import stat as _stat
    if _stat.S_ISLNK(st.st_mode) or not _stat.S_ISDIR(st.st_mode):
        raise PermissionError(f"rendezvous directory {path} is not a plain directory")
    if hasattr(os, "geteuid") and st.st_uid != os.geteuid():
        raise PermissionError(f"rendezvous directory {path} belongs to uid {st.st_uid}, not to this user")
    if not parent and st.st_mode & 0o022:
        raise PermissionError(f"rendezvous directory {path} is writable by others (mode {oct(st.st_mode & 0o777)})")
    if parent and (st.st_mode & 0o002) and not (st.st_mode & _stat.S_ISVTX):
        raise PermissionError(f"rendezvous directory {path} is world-writable without the sticky bit "
                              f"(mode {oct(st.st_mode & 0o7777)})")


def rank_env(environ=None) -> Optional[RankEnv]:
    """The launcher's view of this process, or None when it was started plainly (no RANK / WORLD_SIZE)."""
    e = os.environ if environ is None else environ
    if "RANK" not in e or "WORLD_SIZE" not in e:
        return None
    rank, world = int(e["RANK"]), int(e["WORLD_SIZE"])
    local = int(e.get("LOCAL_RANK", rank))
    d = e.get("GPX_RDZV_DIR")
    owns = e.get("GPX_RDZV_OWNS") == "1"
    if not d:
        # all ranks of one launch share their parent (the launcher's agent process) and its rendezvous port
        d = os.path.join(tempfile.gettempdir(), f"gpx_rdzv_{os.getppid()}_{e.get('MASTER_PORT', '0')}")
        owns = True
    return RankEnv(rank, world, local, d, int(e.get("GPX_RDZV_ATTEMPT", "0")), owns, launch_token(e))


class FileStore:
    """Key -> bytes in a directory; set() publishes atomically, get() polls until the key exists."""

    def __init__(self, directory: str, fresh_after: Optional[float] = None):
        self.dir = directory
        _private_dir(directory)
        # keys older than this are leftovers of an earlier launch that happened to reuse the directory name
        self.fresh_after = fresh_after

    def _path(self, key: str) -> str:
        return os.path.join(self.dir, key)

    def set(self, key: str, value: bytes) -> None:
        tmp = self._path(key) + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(value)
        os.replace(tmp, self._path(key))

    def get(self, key: str, timeout: float = 120.0) -> bytes:
        p = self._path(key)
        t0 = time.monotonic()
        while True:
            try:
                if self.fresh_after is None or os.path.getmtime(p) >= self.fresh_after:
                    with open(p, "rb") as f:
                        return f.read()
            except FileNotFoundError:
                pass
            if time.monotonic() - t0 > timeout:
                raise TimeoutError(f"rendezvous: no '{key}' in {self.dir} after {timeout:.0f} s")
            time.sleep(0.002 if time.monotonic() - t0 < 0.5 else 0.02)

    def gather(self, prefix: str, world: int, timeout: float = 120.0) -> List[bytes]:
        return [self.get(f"{prefix}.{r}", timeout) for r in range(world)]


def _reexec_with_file_transport(env: RankEnv, why: str):
    """Replace this process (same PID: the launcher keeps watching it) by a fresh copy that skips RCCL."""
    sys.stderr.write(f"[gpax_amd.launch] rank {env.rank}: {why}; restarting with the file transport\n")
    sys.stderr.flush()
    os.environ["GPX_RANK_TRANSPORT"] = "file"
    os.environ["GPX_RDZV_ATTEMPT"] = str(env.attempt + 1)
    os.environ["GPX_RDZV_DIR"] = env.rdzv_dir
    if env.token:
        os.environ["GPX_RDZV_TOKEN"] = env.token
    if env.owns_dir:
        os.environ["GPX_RDZV_OWNS"] = "1"
    os.execv(sys.executable, [sys.executable] + sys.argv)


def init_rank(env: RankEnv, device: Optional[int] = None, inflight: Optional[int] = None, transport: Optional[str] = None,
              timeout: float = 120.0, reexec_on_hang: bool = False, make_rank=None, make_uid=None, on_hang=None):
    """Collective: every rank of the launch calls this once and gets its `_lib.Rank` (or whatever `make_rank` builds).

    transport: "rccl", "file" or None = $GPX_RANK_TRANSPORT or "auto" (RCCL; the file transport when RCCL fails on any
    rank — agreed on through the store, so all ranks end up on the same one).
    reexec_on_hang: when the initialisation does not finish within `timeout` seconds, re-execute the process with the
    file transport instead of exiting (launch-script use: bench.py).
    on_hang(why): called (from the watchdog thread) instead of leaving with exit code 70 when the initialisation hangs
    and no re-execution applies — for a caller that has something to report before it leaves (bench.py); it should not
    return.
    make_rank(device, rank, world, unique_id, file_dir, inflight) / make_uid(): injection points for the tests.
    """
    if device is None:  # LOCAL_RANK names the GPU, unless the launcher narrowed the visible devices per rank
        device = env.local_rank
        if make_rank is None:
            from . import _lib as _l
            n_vis = max(1, _l.visible_device_count())
            device = device if device < n_vis else device % n_vis
    device = int(device)
    if make_rank is None or make_uid is None:
        from . import _lib
        make_rank = make_rank or (lambda dev, r, w, uid, fdir, infl: _lib.Rank(dev, r, w, unique_id=uid, file_dir=fdir,
                                                                              inflight=infl))
        make_uid = make_uid or _lib.rccl_unique_id
    transport = (transport or os.environ.get("GPX_RANK_TRANSPORT") or "auto").lower()
    if transport not in ("auto", "rccl", "file"):
        raise ValueError(f"transport {transport!r}")

    t_start = time.time()
    # <directory>/<launch token>/attempt<k>: nothing an earlier launch left in the directory is visible from here, neither
    # store keys nor the file transport's transfer files (their names restart at sequence 0 in every process)
    _private_dir(env.rdzv_dir, parent=True)
    base = os.path.join(env.rdzv_dir, env.token) if env.token else env.rdzv_dir
    _private_dir(base)
    store = FileStore(os.path.join(base, f"attempt{env.attempt}"), fresh_after=t_start - 300.0)
    xfer_dir = os.path.join(store.dir, "xfer")
    _private_dir(xfer_dir)

    done = threading.Event()

    def watchdog():
        if done.wait(timeout):
            return
        if reexec_on_hang and transport == "auto":
            _reexec_with_file_transport(env, f"initialisation did not finish within {timeout:.0f} s")
        sys.stderr.write(f"[gpax_amd.launch] rank {env.rank}: initialisation hung for {timeout:.0f} s; giving up\n")
        sys.stderr.flush()
        if on_hang is not None:
            on_hang(f"communicator initialisation hung for {timeout:.0f} s")
        os._exit(70)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        rk = None
        if transport in ("auto", "rccl"):
            if env.rank == 0:
                try:
                    uid = make_uid()
                except Exception as ex:  # RCCL missing / broken on rank 0: tell everybody
                    uid = b"FAIL:" + str(ex).encode()
                store.set("uid", uid)
            uid = store.get("uid", timeout)
            ok, msg = True, b"ok"
            if uid.startswith(b"FAIL:"):
                ok, msg = False, uid
            else:
                try:
                    rk = make_rank(device, env.rank, env.world, uid, None, inflight)
                except Exception as ex:
                    ok, msg = False, b"FAIL:" + str(ex).encode()
            store.set(f"init.{env.rank}", msg)
            states = store.gather("init", env.world, timeout)
            if not all(s == b"ok" for s in states):
                if rk is not None:
                    rk.close()
                    rk = None
                bad = [f"rank {r}: {s.decode(errors='replace')}" for r, s in enumerate(states) if s != b"ok"]
                if transport == "rccl":
                    raise RuntimeError("RCCL initialisation failed: " + "; ".join(bad))
                if env.rank == 0:
                    sys.stderr.write("[gpax_amd.launch] RCCL unavailable (" + "; ".join(bad) + "): file transport\n")
        if rk is None:
            rk = make_rank(device, env.rank, env.world, None, xfer_dir, inflight)
            store.set(f"finit.{env.rank}", b"ok")
            store.gather("finit", env.world, timeout)
        # the ranks of a sweep work off static blocks: size them by the GPUs' measured speeds (collective probe + one
        # all-reduce, gpx_rank_calibrate; GPX_RANK_WEIGHTED=0: equal blocks)
        if env.world > 1 and os.environ.get("GPX_RANK_WEIGHTED", "1") != "0" and hasattr(rk, "calibrate"):
            rk.speeds = rk.calibrate()
        return rk
    finally:
        done.set()


_default_rank = None
_default_env: Optional[RankEnv] = None


def default_rank():
    """Process-wide `_lib.Rank` of this launch, created on first use from the launcher's environment (collective: every
    rank of the launch must reach its first use).  A process started without a launcher is a world of one."""
    global _default_rank, _default_env
    if _default_rank is None:
        env = rank_env()
        if env is None:
            env = RankEnv(0, 1, int(os.environ.get("GPX_DEVICE", "0")), tempfile.mkdtemp(prefix="gpx_rdzv_"), 0, True)
        _default_env = env
        _default_rank = init_rank(env)
    return _default_rank


def shutdown() -> None:
    """Collective end of the process-wide rank (barrier, close, rendezvous directory removed)."""
    global _default_rank, _default_env
    if _default_rank is not None:
        finalize(_default_env, _default_rank)
        _default_rank = _default_env = None


def finalize(env: RankEnv, rk=None) -> None:
    """Collective end of a run: barrier, close, and rank 0 removes a rendezvous directory this module derived."""
    if rk is not None:
        try:
            rk.barrier()
        finally:
            rk.close()
    if env.rank == 0 and env.owns_dir:
        shutil.rmtree(env.rdzv_dir, ignore_errors=True)


def spawn_command(script: str, argv: List[str], nranks: int) -> dict:
    """What `spawn_ranks` runs: the command line of every rank and the per-rank environment it adds."""
    return {"argv": [sys.executable, os.path.abspath(script)] + list(argv),
            "env_per_rank": {"RANK": "<r>", "LOCAL_RANK": "<r>", "WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1",
                             "GPX_RDZV_DIR": "<fresh temporary directory>", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
            "nranks": nranks, "launcher": "gpax_amd.launch.spawn_ranks (subprocess, no torch)"}


def spawn_ranks(script: str, argv: List[str], nranks: int, timeout: Optional[float] = None) -> int:
    """Minimal single-node launcher: N copies of `python script argv...`, one per rank, sharing a fresh rendezvous
    directory.  Returns the first non-zero exit code (the remaining ranks are terminated), else 0."""
    rdzv = tempfile.mkdtemp(prefix="gpx_rdzv_")
    token = "l" + os.urandom(8).hex()
    procs = []
    try:
        for r in range(nranks):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1",
                        "GPX_RDZV_DIR": rdzv, "GPX_RDZV_TOKEN": token})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
        t0 = time.monotonic()
        rc = 0
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                for p in live:  # exactly the processes started above
                    p.terminate()
                for p in live:
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
        return rc
    finally:
        shutil.rmtree(rdzv, ignore_errors=True)
