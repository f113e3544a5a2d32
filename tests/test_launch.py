"""CPU: the torch-free one-process-per-GPU plumbing (gpax_amd/launch.py) with world sizes 2 and 3 — the rendezvous
store, the unique-id hand-over, the all-ranks agreement on the transport (RCCL fails on one rank => every rank falls
back to the file transport together), the sharding rule of the library (gpx_shard_range, host only) and the
spawn launcher.  The communicator itself needs a GPU: tests/test_gpu_rank.py."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeRank:
    """Stands in for _lib.Rank: records how it was built."""

    def __init__(self, dev, rank, world, uid, fdir, infl):
        self.args = dict(device=dev, rank=rank, world=world, uid=uid, file_dir=fdir, inflight=infl)
        self.closed = False

    def close(self):
        self.closed = True

    def barrier(self):
        pass


def _worker(rank, world, rdzv, mode, q):
    sys.path.insert(0, ROOT)
    from gpax_amd import launch
    env = launch.RankEnv(rank, world, rank, rdzv)

    def make_uid():
        if mode == "uid_fails":
            raise RuntimeError("no librccl")
        return bytes([7]) * 128

    def make_rank(dev, r, w, uid, fdir, infl):
        if mode == "rank1_fails" and r == 1 and fdir is None:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        return FakeRank(dev, r, w, uid, fdir, infl)

    try:
        rk = launch.init_rank(env, inflight=2, transport="rccl" if mode == "rccl_only" else "auto", timeout=30.0,
                              make_rank=make_rank, make_uid=make_uid)
        q.put((rank, "ok", rk.args))
    except Exception as ex:
        q.put((rank, "error", str(ex)))


def _run(world, mode, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, str(tmp_path / "rdzv"), mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    return sorted(out)


@pytest.mark.parametrize("world", [2, 3])
def test_every_rank_gets_rank_0s_unique_id(world, tmp_path):
    out = _run(world, "fine", tmp_path)
    assert [o[1] for o in out] == ["ok"] * world
    for r, _, args in out:
        assert args["uid"] == bytes([7]) * 128 and args["file_dir"] is None
        assert args["rank"] == r and args["world"] == world and args["device"] == r and args["inflight"] == 2


@pytest.mark.parametrize("mode", ["rank1_fails", "uid_fails"])
def test_a_failure_on_one_rank_moves_all_ranks_to_the_file_transport(mode, tmp_path):
    out = _run(3, mode, tmp_path)
    assert [o[1] for o in out] == ["ok"] * 3
    dirs = {o[2]["file_dir"] for o in out}
    assert len(dirs) == 1 and None not in dirs and os.path.isdir(dirs.pop())
    assert all(o[2]["uid"] is None for o in out)


def test_transport_rccl_does_not_fall_back(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # rccl demanded, rank 1 fails: every rank raises instead of changing transport
    procs = [ctx.Process(target=_worker_rccl_only, args=(r, 2, str(tmp_path / "rdzv"), q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(30)
    assert [o[1] for o in out] == ["error", "error"] and "rank 1" in out[0][2]


def _worker_rccl_only(rank, world, rdzv, q):
    sys.path.insert(0, ROOT)
    from gpax_amd import launch
    env = launch.RankEnv(rank, world, rank, rdzv)

    def make_rank(dev, r, w, uid, fdir, infl):
        if r == 1:
            raise RuntimeError("boom")
        return FakeRank(dev, r, w, uid, fdir, infl)

    try:
        launch.init_rank(env, transport="rccl", timeout=30.0, make_rank=make_rank, make_uid=lambda: b"x" * 128)
        q.put((rank, "ok", ""))
    except Exception as ex:
        q.put((rank, "error", str(ex)))


def test_file_store_ignores_leftovers_of_an_earlier_launch(tmp_path):
    from gpax_amd.launch import FileStore
    old = FileStore(str(tmp_path))
    old.set("uid", b"stale")
    past = time.time() - 3600
    os.utime(tmp_path / "uid", (past, past))
    st = FileStore(str(tmp_path), fresh_after=time.time() - 300)
    with pytest.raises(TimeoutError):
        st.get("uid", timeout=0.2)
    st.set("uid", b"fresh")
    assert st.get("uid", timeout=1.0) == b"fresh"


def test_a_reused_directory_never_shows_an_earlier_launch(tmp_path):
    """the file transport's transfer files are named <tag>.<seq>.<rank> with seq restarting at 0 in every
    process, and FileStore's freshness window lets a 5-minute-old key through.  Two launches that share a rendezvous
    directory (GPX_RDZV_DIR set by hand and run twice, a crashed run, a launcher restarting its workers) must still not
    see each other: each works under <directory>/<launch token>/, and the token differs between launches."""
    from gpax_amd import launch
    rdzv = str(tmp_path / "rdzv")
    first = launch.rank_env({"RANK": "0", "WORLD_SIZE": "1", "GPX_RDZV_DIR": rdzv, "GPX_RDZV_TOKEN": "launchA"})
    second = launch.rank_env({"RANK": "0", "WORLD_SIZE": "1", "GPX_RDZV_DIR": rdzv, "GPX_RDZV_TOKEN": "launchB"})
    assert first.token == "launchA" and second.token == "launchB"
    rk1 = launch.init_rank(first, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    d1 = rk1.args["file_dir"]
    with open(os.path.join(d1, "in.0.0"), "wb") as f:  # what a first launch's sweep leaves behind
        f.write(b"stale payload")
    rk2 = launch.init_rank(second, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    d2 = rk2.args["file_dir"]
    assert d1 != d2 and os.listdir(d2) == [] and os.path.exists(os.path.join(d1, "in.0.0"))
    # without a token from the launcher: the common parent of the ranks by PID and start time — the same for every
    # rank of one launch, and a sanitised path component
    t = launch.launch_token({"MASTER_PORT": "29511", "TORCHELASTIC_RUN_ID": "run/../7"})
    assert t == launch.launch_token({"MASTER_PORT": "29511", "TORCHELASTIC_RUN_ID": "run/../7"})
    assert t.startswith(f"p{os.getppid()}_") and "/" not in t and ".." not in t
    assert launch.launch_token({"MASTER_PORT": "29512"}) != t
    # an elastic agent restarting its workers keeps PID, port and run id — the restart count tells the
    # incarnations apart
    base = {"MASTER_PORT": "29511", "TORCHELASTIC_RUN_ID": "r7"}
    assert launch.launch_token(dict(base, TORCHELASTIC_RESTART_COUNT="1")) != launch.launch_token(base)
    assert launch.launch_token(dict(base, TORCHELASTIC_RESTART_COUNT="0")) == launch.launch_token(base)


def test_a_hand_made_rendezvous_directory_with_group_write_is_accepted_but_its_subdirectories_are_private(tmp_path):
    """GPX_RDZV_DIR made by hand under umask 002 is 0775; only owner and not-a-symlink are checked on the
    directory the user hands over — the ids live in the <token>/attempt subdirectories the module creates (0700)."""
    from gpax_amd import launch
    rdzv = tmp_path / "by_hand"
    rdzv.mkdir()
    os.chmod(rdzv, 0o775)
    env = launch.rank_env({"RANK": "0", "WORLD_SIZE": "1", "GPX_RDZV_DIR": str(rdzv), "GPX_RDZV_TOKEN": "t"})
    rk = launch.init_rank(env, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    fdir = rk.args["file_dir"]
    assert fdir.startswith(str(rdzv)) and (os.stat(fdir).st_mode & 0o777) == 0o700
    assert (os.stat(os.path.join(str(rdzv), "t")).st_mode & 0o777) == 0o700
    link = tmp_path / "link"
    os.symlink(rdzv, link)
    env2 = launch.rank_env({"RANK": "0", "WORLD_SIZE": "1", "GPX_RDZV_DIR": str(link), "GPX_RDZV_TOKEN": "t"})
    with pytest.raises(PermissionError, match="not a plain directory"):
        launch.init_rank(env2, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    # world-writable without the sticky bit is refused (anybody could rename the <token> subdirectory between the
    # checks and plant their own); with it — /tmp's mode — only the owner of an entry can rename it
    open_dir = tmp_path / "open"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    env3 = launch.rank_env({"RANK": "0", "WORLD_SIZE": "1", "GPX_RDZV_DIR": str(open_dir), "GPX_RDZV_TOKEN": "t"})
    with pytest.raises(PermissionError, match="world-writable without the sticky bit"):
        launch.init_rank(env3, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    os.chmod(open_dir, 0o1777)
    rk3 = launch.init_rank(env3, transport="file", timeout=10.0, make_rank=FakeRank, make_uid=lambda: b"x" * 128)
    assert (os.stat(rk3.args["file_dir"]).st_mode & 0o777) == 0o700


def test_the_rendezvous_directory_must_be_private(tmp_path):
    """a directory another user could have planted (a symbolic link, group / world writable) is refused —
    it carries the RCCL bootstrap id — and what the store creates is mode 0700."""
    from gpax_amd.launch import FileStore
    st = FileStore(str(tmp_path / "fresh"))
    assert (os.stat(st.dir).st_mode & 0o777) == 0o700
    loose = tmp_path / "loose"
    loose.mkdir()
    os.chmod(loose, 0o777)
    with pytest.raises(PermissionError, match="writable by others"):
        FileStore(str(loose))
    target = tmp_path / "elsewhere"
    target.mkdir(mode=0o700)
    link = tmp_path / "link"
    os.symlink(target, link)
    with pytest.raises(PermissionError, match="not a plain directory"):
        FileStore(str(link))


def test_rank_env_reads_the_launcher_variables():
    from gpax_amd.launch import rank_env
    assert rank_env({}) is None
    e = rank_env({"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3", "MASTER_PORT": "29511"})
    assert (e.rank, e.world, e.local_rank) == (3, 8, 3) and e.owns_dir and e.rdzv_dir.endswith(f"_{os.getppid()}_29511")
    e = rank_env({"RANK": "1", "WORLD_SIZE": "2", "GPX_RDZV_DIR": "/x/y", "GPX_RDZV_ATTEMPT": "1"})
    assert e.rdzv_dir == "/x/y" and not e.owns_dir and e.attempt == 1 and e.local_rank == 1


def test_the_librarys_sharding_rule():
    """gpx_shard_range is host-only code of libgpx: contiguous blocks, sizes differing by at most one, every sample once."""
    from gpax_amd import _lib
    for S in (0, 1, 7, 8, 1000):
        for parts in (1, 2, 3, 8):
            blocks = [_lib.shard_range(S, r, parts) for r in range(parts)]
            assert blocks[0][0] == 0 and blocks[-1][1] == S
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(parts - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_spawn_ranks_sets_the_rank_environment_and_reports_failures(tmp_path):
    from gpax_amd import launch
    script = tmp_path / "probe.py"
    script.write_text(
        "import os, sys, json\n"
        "r = os.environ['RANK']\n"
        "open(os.path.join(sys.argv[1], 'r' + r), 'w').write(json.dumps({k: os.environ[k] for k in "
        "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'GPX_RDZV_DIR', 'MASTER_ADDR')}))\n"
        "sys.exit(3 if (len(sys.argv) > 2 and r == '1') else 0)\n")
    assert launch.spawn_ranks(str(script), [str(tmp_path)], 3) == 0
    recs = [json.loads((tmp_path / f"r{r}").read_text()) for r in range(3)]
    assert [x["RANK"] for x in recs] == ["0", "1", "2"] and {x["WORLD_SIZE"] for x in recs} == {"3"}
    assert len({x["GPX_RDZV_DIR"] for x in recs}) == 1 and recs[0]["MASTER_ADDR"] == "127.0.0.1"
    assert not os.path.exists(recs[0]["GPX_RDZV_DIR"])  # the launcher removes its rendezvous directory
    assert launch.spawn_ranks(str(script), [str(tmp_path), "fail"], 2) == 3


def test_a_hung_initialisation_reexecutes_the_rank_with_the_file_transport(tmp_path):
    """A peer that dies inside ncclCommInitRank leaves the others blocked in a C call nothing can interrupt: the
    watchdog replaces the process (same PID, so the launcher keeps watching it) by a copy that skips RCCL."""
    from gpax_amd import launch
    script = tmp_path / "hang.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from gpax_amd import launch\n"
        "env = launch.rank_env()\n"
        "class R:\n"
        "    def __init__(self, fdir): self.fdir = fdir\n"
        "    def close(self): pass\n"
        "def make_rank(dev, r, w, uid, fdir, infl):\n"
        "    if fdir is None:\n"
        "        time.sleep(3600)  # 'ncclCommInitRank' never returns\n"
        "    return R(fdir)\n"
        "rk = launch.init_rank(env, timeout=1.5, reexec_on_hang=True, make_rank=make_rank, make_uid=lambda: b'u' * 128)\n"
        "open(os.path.join(sys.argv[1], 'done%d' % env.rank), 'w').write('%s %d %s' % (os.environ.get('GPX_RANK_TRANSPORT'), "
        "env.attempt, rk.fdir))\n")
    assert launch.spawn_ranks(str(script), [str(tmp_path)], 2, timeout=120) == 0
    for r in range(2):
        transport, attempt, fdir = (tmp_path / f"done{r}").read_text().split()
        assert transport == "file" and attempt == "1" and fdir.endswith(os.path.join("attempt1", "xfer"))


def test_a_hung_initialisation_hands_control_to_the_callers_on_hang(tmp_path):
    """No re-execution applies (the transport is fixed): a caller with something to report — bench.py and its replicas
    leg — is called back from the watchdog thread instead of the process leaving with exit code 70."""
    from gpax_amd import launch
    script = tmp_path / "hang2.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from gpax_amd import launch\n"
        "env = launch.rank_env()\n"
        "def make_rank(dev, r, w, uid, fdir, infl):\n"
        "    time.sleep(3600)\n"
        "def on_hang(why):\n"
        "    open(os.path.join(sys.argv[1], 'hang%d' % env.rank), 'w').write(why)\n"
        "    os._exit(0)\n"
        "launch.init_rank(env, timeout=1.0, transport='file', make_rank=make_rank, make_uid=lambda: b'u' * 128, on_hang=on_hang)\n")
    assert launch.spawn_ranks(str(script), [str(tmp_path)], 2, timeout=120) == 0
    for r in range(2):
        assert "hung for 1 s" in (tmp_path / f"hang{r}").read_text()


def test_weighted_shard_ranges_follow_the_weights_and_partition_exactly():
    from gpax_amd import _lib
    """gpx_shard_ranges_weighted (host only): the ranks' blocks of a calibrated sweep (gpx_rank_calibrate) — contiguous, every
    sample once, sizes within one of S w / sum(w), equal weights = gpx_shard_range, bad weights fall back to it."""
    rng = np.random.default_rng(0)
    for S in (0, 1, 7, 40, 1000, 1001):
        for parts in (1, 2, 3, 8):
            eq = _lib.shard_ranges_weighted(S, np.ones(parts))
            assert eq == [_lib.shard_range(S, r, parts) for r in range(parts)]
            w = rng.uniform(0.85, 1.15, parts)
            blocks = _lib.shard_ranges_weighted(S, w)
            assert blocks[0][0] == 0 and blocks[-1][1] == S
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(parts - 1))
            for r, (lo, hi) in enumerate(blocks):
                assert abs((hi - lo) - S * w[r] / w.sum()) < 1.0
            bad = w.copy()
            bad[0] = 0.0
            assert _lib.shard_ranges_weighted(S, bad) == eq
    # a GPU 8 % slower than seven equal ones gets the smaller block of C4's 1000 samples
    sizes = [hi - lo for lo, hi in _lib.shard_ranges_weighted(1000, [1.0] * 7 + [0.92])]
    assert sizes[-1] == min(sizes) and sizes[-1] in (116, 117) and sum(sizes) == 1000
