"""The C++ host mirror (nerf-cuda_amd/host): snapshot parsing and config derivation on the CPU,
the testbed and the render_server wire protocol on the GPU."""
import json
import os
import socket
import subprocess
import time
from pathlib import Path

import numpy as np
import pytest

import models
import nerfhip as nh
import synthetic as syn

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "nerf-cuda_amd" / "host"
# the QUIT message (statistics + shutdown) is a test hook the server only honours with this variable set
SERVER_TEST_ENV = dict(os.environ, NRF_SERVER_TEST_HOOKS="1", NRF_SERVER_BIND="127.0.0.1")


@pytest.fixture(scope="module")
def snapshot(tmp_path_factory):
    d = tmp_path_factory.mktemp("snap")
    desc, keep, cfg = models.build_model(log2_hashmap_size=12, H=32)
    path = d / "tiny.msgpack"
    syn.write_snapshot(path, cfg, keep[0], keep[1])
    return path, desc, keep, cfg


def _oracle_rgb8(desc, cam, pose, W, H):
    """The reference's answer to one pose request (render_server.cu:93-101: render_frame -> Image.rgb): the CPU
    oracle's frame (per-ray schedule), quantised like nerf_render.cu:352-359."""
    import oracle_py as op
    rgba, depth, _ = op.Oracle(desc).render(cam, pose, W, H, schedule=op.SCHED_PER_RAY)
    return op.quantize_u8(rgba, depth)[0]


def _assert_within_one_lsb(got_u8, want_u8, what):
    d = np.abs(got_u8.astype(np.int16) - want_u8.astype(np.int16))
    assert d.max() <= 1, f"{what}: server bytes differ from the oracle's quantised frame by {d.max()} LSB"
    assert (d > 0).mean() < 0.02, f"{what}: {100 * (d > 0).mean():.2f} % of the bytes differ"  # rounding-edge cases only


def _info(path):
    r = subprocess.run([str(HOST / "snapshot_info"), str(path)], capture_output=True, text=True)
    return r


def test_cpp_snapshot_parsing_matches_python(snapshot):
    path, desc, keep, cfg = snapshot
    r = _info(path)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["rc"] == 0 and d["n_params"] == desc.n_params == d["expected"]
    assert d["n_grid"] == desc.n_density_grid
    for k, v in dict(grid_type=desc.grid_type, n_levels=desc.n_levels, F=desc.n_features_per_level,
                     log2T=desc.log2_hashmap_size, base=desc.base_resolution, n_neurons=desc.n_neurons,
                     dh=desc.density_hidden_layers, da=desc.density_activation, doa=desc.density_output_activation,
                     dno=desc.density_n_output, sa=desc.sigma_activation, rh=desc.rgb_hidden_layers,
                     ra=desc.rgb_activation, roa=desc.rgb_output_activation, dir=desc.dir_encoding,
                     shdeg=desc.sh_degree, cascade=desc.cascade, H=desc.density_grid_size).items():
        assert d[k] == v, k
    assert d["pls"] == pytest.approx(desc.per_level_scale, rel=1e-7)
    assert d["bound"] == desc.bound and d["scale"] == pytest.approx(desc.scale)
    assert d["mean_density"] == pytest.approx(desc.mean_density, rel=1e-6)
    # python round trip of the same file
    back = syn.read_snapshot(path)
    d2, _ = nh.desc_from_config(back)
    assert d2.n_params == desc.n_params and d2.per_level_scale == desc.per_level_scale


def test_cpp_error_behaviour(tmp_path, snapshot):
    # nerf_render.cu:73-76: missing file
    r = _info(tmp_path / "nope.msgpack")
    assert r.returncode == 1 and "does not exist" in r.stderr
    # nerf_render.cu:434-436: no snapshot block
    import msgpack
    p = tmp_path / "nosnap.msgpack"
    p.write_bytes(msgpack.packb({"encoding": {"otype": "HashGrid"}}))
    r = _info(p)
    assert r.returncode == 1 and "does not contain a snapshot" in r.stderr
    # nerf_render.cu:467-469: density grid size vs cascade
    path, desc, keep, cfg = snapshot
    bad = dict(cfg)
    bad["snapshot"] = dict(cfg["snapshot"], cascade=2)
    p = tmp_path / "badgrid.msgpack"
    syn.write_snapshot(p, bad, keep[0], keep[1])
    r = _info(p)
    assert r.returncode == 1 and "Incompatible number of grid cascades" in r.stderr


@pytest.mark.gpu
def test_testbed_matches_python_binding(tmp_path, snapshot):
    path, desc, keep, cfg = snapshot
    W, H = 120, 88
    r = subprocess.run([str(HOST / "testbed"), str(path), str(W), str(H), str(tmp_path) + "/"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Process time" in r.stdout
    got = np.fromfile(tmp_path / "image.rgb", np.uint8).reshape(H, W, 3)
    assert (tmp_path / "image.png").stat().st_size > W * H * 3 and (tmp_path / "deep.png").exists()
    assert (tmp_path / "tonemapped.png").stat().st_size > W * H * 3  # render buffer chain ran (host u8 -> tonemap)
    ctx = nh.NerfHip(0)
    ctx.load_model(desc)
    ctx.set_resolution(W, H)
    s = np.float32(W) / np.float32(500.0)
    cam = np.array([3550.115 / 8, 3554.515 / 8, 3010.45 / 8, 1996.027 / 8], np.float32) * s
    ctx.render(cam, syn.REFERENCE_MAIN_POSE)
    rgb8, _ = ctx.read_u8()
    np.testing.assert_array_equal(got, rgb8)
    assert rgb8.min() < 250  # the object is in view
    ctx.close()
    # the NGPU form of the class (nrf_group underneath): three members, all on the one device of this box
    import os
    out3 = tmp_path / "n3"
    out3.mkdir()
    env = dict(os.environ, NERF_NGPU="3", NERF_DEVICES="0,0,0")
    r = subprocess.run([str(HOST / "testbed"), str(path), str(W), str(H), str(out3) + "/"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    np.testing.assert_array_equal(np.fromfile(out3 / "image.rgb", np.uint8).reshape(H, W, 3), rgb8)
    # the same binary with the exchange step on RCCL (NRF_GROUP_GATHER=rccl: nrf_group_create opens librccl, builds a
    # one-rank communicator and the lone member renders its shard tile-major, sends it to itself, untiles): same bytes; and a
    # group that lists a device twice cannot take that transport -- the error is the library's, loud, not a silent fallback
    outr = tmp_path / "rccl"
    outr.mkdir()
    r = subprocess.run([str(HOST / "testbed"), str(path), str(W), str(H), str(outr) + "/"], capture_output=True, text=True, timeout=180,
                       env=dict(os.environ, NRF_GROUP_GATHER="rccl"))
    assert r.returncode == 0, r.stderr + r.stdout
    np.testing.assert_array_equal(np.fromfile(outr / "image.rgb", np.uint8).reshape(H, W, 3), rgb8)
    r = subprocess.run([str(HOST / "testbed"), str(path), str(W), str(H), str(outr) + "/"], capture_output=True, text=True, timeout=180,
                       env=dict(os.environ, NRF_GROUP_GATHER="rccl", NERF_NGPU="2", NERF_DEVICES="0,0"))
    assert r.returncode != 0 and "DISTINCT" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_render_server_wire_protocol(snapshot):
    path, desc, keep, cfg = snapshot
    W, H, port = 64, 64, 23457
    srv = subprocess.Popen([str(HOST / "render_server"), str(port), str(path), str(W), str(H)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=SERVER_TEST_ENV)
    try:
        sock = None
        for _ in range(200):
            try:
                sock = socket.create_connection(("127.0.0.1", port), timeout=1.0)
                break
            except OSError:
                time.sleep(0.1)
        assert sock is not None, "server did not come up"
        sock.settimeout(30)
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        for pose in (syn.REFERENCE_MAIN_POSE, syn.orbit_pose(200, 20)):
            sock.sendall(np.ascontiguousarray(pose, np.float32).tobytes())  # 64 bytes, row-major 4x4
            buf = bytearray()
            while len(buf) < 3 * W * H:
                chunk = sock.recv(3 * W * H - len(buf))
                assert chunk, "connection closed early"
                buf += chunk
            ctx.render(cam, pose)
            rgb8, _ = ctx.read_u8()
            got = np.frombuffer(bytes(buf), np.uint8).reshape(H, W, 3)
            np.testing.assert_array_equal(got, rgb8)
            _assert_within_one_lsb(got, _oracle_rgb8(desc, cam, pose, W, H), "raw 64-byte pose")  # HIP vs the oracle
            assert got.min() < 250  # the object is in view: the comparison is not background against background
        # extended request on the same connection: "NRF1", u32 n, n x {cam[4], pose[16]}: one launch, per-view intrinsics
        views = [(cam * np.float32(1.0 + 0.1 * i), syn.orbit_pose(70.0 * i, 15.0 + 5 * i)) for i in range(3)]
        msg = b"NRF1" + np.uint32(len(views)).tobytes()
        for c, p in views:
            msg += np.ascontiguousarray(c, np.float32).tobytes() + np.ascontiguousarray(p, np.float32).tobytes()
        sock.sendall(msg)
        for c, p in views:
            buf = bytearray()
            while len(buf) < 3 * W * H:
                chunk = sock.recv(3 * W * H - len(buf))
                assert chunk, "connection closed early"
                buf += chunk
            ctx.render(c, p)
            got = np.frombuffer(bytes(buf), np.uint8).reshape(H, W, 3)
            np.testing.assert_array_equal(got, ctx.read_u8()[0])
            _assert_within_one_lsb(got, _oracle_rgb8(desc, c, p, W, H), "NRF1 request")
        quit_msg = np.zeros(16, np.float32)
        quit_msg[:1] = np.frombuffer(b"QUIT", np.float32)
        sock.sendall(quit_msg.tobytes())
        sock.close()
        srv.wait(timeout=20)
        ctx.close()
    finally:
        if srv.poll() is None:
            srv.kill()


@pytest.mark.gpu
def test_render_server_batches_concurrent_clients(snapshot):
    """BASELINE config 5 shape: several clients connected at once, every one pipelining requests.  The
    server folds whatever is queued into one nrf_render_views launch; every answer must equal the
    single-request render of that pose, and fewer launches than frames must have been used."""
    import threading

    path, desc, keep, cfg = snapshot
    W, H, port = 96, 64, 23459
    srv = subprocess.Popen([str(HOST / "render_server"), str(port), str(path), str(W), str(H)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=SERVER_TEST_ENV)
    n_clients, per_client = 6, 5
    try:
        socks = []
        for c in range(n_clients):
            s = None
            for _ in range(200):
                try:
                    s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
                    break
                except OSError:
                    time.sleep(0.1)
            assert s is not None, "server did not come up"
            s.settimeout(60)
            socks.append(s)
        poses = [[syn.orbit_pose(37.0 * (c * per_client + i), -10.0 + 9.0 * i) for i in range(per_client)]
                 for c in range(n_clients)]
        got = [[None] * per_client for _ in range(n_clients)]
        errors = []

        def client(c):
            try:
                for i in range(per_client):
                    socks[c].sendall(np.ascontiguousarray(poses[c][i], np.float32).tobytes())
                    buf = bytearray()
                    while len(buf) < 3 * W * H:
                        chunk = socks[c].recv(3 * W * H - len(buf))
                        if not chunk:
                            raise RuntimeError("connection closed early")
                        buf += chunk
                    got[c][i] = np.frombuffer(bytes(buf), np.uint8).reshape(H, W, 3)
            except Exception as e:  # noqa: BLE001
                errors.append((c, repr(e)))

        threads = [threading.Thread(target=client, args=(c,)) for c in range(n_clients)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        for c in range(n_clients):
            for i in range(per_client):
                ctx.render(cam, poses[c][i])
                np.testing.assert_array_equal(got[c][i], ctx.read_u8()[0])
                _assert_within_one_lsb(got[c][i], _oracle_rgb8(desc, cam, poses[c][i], W, H), f"client {c} request {i}")
        ctx.close()
        quit_msg = np.zeros(16, np.float32)
        quit_msg[:1] = np.frombuffer(b"QUIT", np.float32)
        socks[0].sendall(quit_msg.tobytes())
        for s in socks:
            s.close()
        out, _ = srv.communicate(timeout=30)
        line = [ln for ln in out.splitlines() if ln.startswith("batches ")][-1].split()
        batches, frames = int(line[1]), int(line[3])
        assert frames == n_clients * per_client and batches < frames, (batches, frames)
    finally:
        if srv.poll() is None:
            srv.kill()


@pytest.mark.gpu
def test_render_server_per_gpu_queues_64_clients(snapshot):
    """BASELINE config 5 shape through the real server: 64 concurrent clients, two per-GPU queues (both workers on the one
    device of this box: NERF_DEVICES=0,0), whole requests dealt to the less loaded queue, batches of up to 64 views, batch
    k + 1 rendering while batch k is copied and sent.  Every reply must equal the binding's render of its pose; a sample
    is compared with the ORACLE's quantised frame (<= 1 LSB); both workers must have rendered."""
    import threading

    path, desc, keep, cfg = snapshot
    W, H, port = 64, 64, 23463
    env = dict(SERVER_TEST_ENV, NERF_DEVICES="0,0")
    srv = subprocess.Popen([str(HOST / "render_server"), str(port), str(path), str(W), str(H)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=env)
    n_clients, per_client = 64, 4
    try:
        socks = []
        for c in range(n_clients):
            s = _connect(port)
            assert s is not None, "server did not come up"
            s.settimeout(120)
            socks.append(s)
        poses = [[syn.orbit_pose(5.625 * c + 90.0 * i, -10.0 + 20.0 * ((c + i) % 4)) for i in range(per_client)] for c in range(n_clients)]
        got = [[None] * per_client for _ in range(n_clients)]
        errors = []

        def client(c):
            try:
                for i in range(per_client):
                    socks[c].sendall(np.ascontiguousarray(poses[c][i], np.float32).tobytes())
                    got[c][i] = np.frombuffer(_recv_exact(socks[c], 3 * W * H), np.uint8).reshape(H, W, 3)
            except Exception as e:  # noqa: BLE001
                errors.append((c, repr(e)))

        threads = [threading.Thread(target=client, args=(c,)) for c in range(n_clients)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(180)
        assert not errors, errors[:3]
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        for c in range(n_clients):
            for i in range(per_client):
                rgb, _ = ctx.render_host_u8([cam], [poses[c][i]])
                np.testing.assert_array_equal(got[c][i], rgb[0], err_msg=f"client {c} request {i}")
        ctx.close()
        for c, i in ((0, 0), (13, 1), (31, 2), (47, 3), (63, 0), (20, 2)):
            _assert_within_one_lsb(got[c][i], _oracle_rgb8(desc, cam, poses[c][i], W, H), f"client {c} request {i}")
        # the STAT hook answers with one 256-byte line of text
        stat_msg = np.zeros(16, np.float32)
        stat_msg[:1] = np.frombuffer(b"STAT", np.float32)
        socks[1].sendall(stat_msg.tobytes())
        words = _recv_exact(socks[1], 256).split(b"\0", 1)[0].decode().split()
        st = {words[i]: float(words[i + 1]) for i in range(0, len(words), 2)}
        assert st["frames"] == n_clients * per_client and st["workers"] == 2 and st["batches"] < st["frames"] and st["gpu_ms"] > 0
        quit_msg = np.zeros(16, np.float32)
        quit_msg[:1] = np.frombuffer(b"QUIT", np.float32)
        socks[0].sendall(quit_msg.tobytes())
        for s in socks:
            s.close()
        out, _ = srv.communicate(timeout=30)
        assert "2 GPU queue(s)" in out
    finally:
        if srv.poll() is None:
            srv.kill()


def test_binary_snapshot_blobs_cpp_and_python(tmp_path, snapshot):
    """`params_binary` / `density_grid_binary` (+ `*_type` "__half" | "float": instant-ngp's convention) carry the
    same values in the same order as the reference's arrays of numbers; the C++ loader and the Python one agree
    on every element (order-dependent checksums), and fp32 blobs reproduce the array form exactly."""
    path, desc, keep, cfg = snapshot
    params, grid = keep

    def checksums(p, g):
        w = (np.arange(p.size, dtype=np.float64) % 97) + 1
        v = (np.arange(g.size, dtype=np.float64) % 89) + 1
        return float((p.astype(np.float64) * w).sum()), float((g.astype(np.float64) * v).sum())

    base = json.loads(_info(path).stdout.strip().splitlines()[-1])
    for kind, cast in (("float", np.float32), ("__half", np.float16)):
        f = tmp_path / f"bin_{kind}.msgpack"
        syn.write_snapshot(f, cfg, params, grid, binary=kind)
        assert f.stat().st_size < path.stat().st_size
        r = _info(f)
        assert r.returncode == 0, r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        want_p, want_g = checksums(params.astype(cast).astype(np.float32), grid.astype(cast).astype(np.float32))
        assert d["n_params"] == params.size and d["n_grid"] == grid.size and d["rc"] == 0
        assert abs(d["psum"] - want_p) <= 1e-9 * abs(want_p) and abs(d["gsum"] - want_g) <= 1e-9 * max(abs(want_g), 1.0)
        if kind == "float":
            assert d["psum"] == base["psum"] and d["gsum"] == base["gsum"]
        d2, keep2 = nh.desc_from_config(syn.read_snapshot(f))
        np.testing.assert_array_equal(keep2[0], params.astype(cast).astype(np.float32))
        np.testing.assert_array_equal(keep2[1], grid.astype(cast).astype(np.float32))
    # unknown element type / odd blob sizes are errors, not guesses
    import msgpack
    bad = syn.read_snapshot(tmp_path / "bin_float.msgpack")
    bad["snapshot"]["params_type"] = "double"
    (tmp_path / "bad.msgpack").write_bytes(msgpack.packb(bad, use_single_float=True, use_bin_type=True))
    r = _info(tmp_path / "bad.msgpack")
    assert r.returncode == 1 and "unknown element type" in r.stderr


@pytest.mark.gpu
def test_python_nerf_render_mirror_matches_cpp_testbed(tmp_path, snapshot):
    """nerfhip.NerfRender (the reference class's method names on top of nrf_group) produces the testbed's image."""
    path, desc, keep, cfg = snapshot
    W, H = 120, 88
    r = subprocess.run([str(HOST / "testbed"), str(path), str(W), str(H), str(tmp_path) + "/"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    want = np.fromfile(tmp_path / "image.rgb", np.uint8).reshape(H, W, 3)
    render = nh.NerfRender(devices=[0, 0])  # two members on the one device of this box
    with pytest.raises(RuntimeError):
        render.load_snapshot(tmp_path / "missing.msgpack")
    render.reload_network_from_file(path)
    render.set_resolution((W, H))
    s = np.float32(W) / np.float32(500.0)
    cam = np.array([3550.115 / 8, 3554.515 / 8, 3010.45 / 8, 1996.027 / 8], np.float32) * s
    rgb, depth = render.render_frame(cam, syn.REFERENCE_MAIN_POSE)
    np.testing.assert_array_equal(rgb, want)
    both = render.render_frames([cam, cam], [syn.REFERENCE_MAIN_POSE, syn.orbit_pose(10, 20)])
    np.testing.assert_array_equal(both[0][0], want)
    assert both[1][0].shape == (H, W, 3) and not np.array_equal(both[1][0], want)
    render.close()


def _connect(port, tries=200):
    for _ in range(tries):
        try:
            return socket.create_connection(("127.0.0.1", port), timeout=1.0)
        except OSError:
            time.sleep(0.1)
    return None


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        assert chunk, "connection closed early"
        buf += chunk
    return bytes(buf)


@pytest.mark.gpu
def test_render_server_survives_rude_clients(snapshot):
    """A client that disconnects in the middle of a reply must not end the server (the reference ignores SIGPIPE
    through sockpp::socket_initializer), a remote "QUIT" is an ordinary (degenerate) pose unless the test hook is
    enabled, and an extended request larger than one launch is served in bounded chunks."""
    path, desc, keep, cfg = snapshot
    W, H, port = 256, 192, 23461
    env = dict(os.environ, NRF_SERVER_BIND="127.0.0.1")
    env.pop("NRF_SERVER_TEST_HOOKS", None)
    srv = subprocess.Popen([str(HOST / "render_server"), str(port), str(path), str(W), str(H)], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, env=env)
    try:
        pose = np.ascontiguousarray(syn.REFERENCE_MAIN_POSE, np.float32).tobytes()
        for _ in range(3):  # pipelined requests, then gone without reading a byte: the replies hit a closed socket
            rude = _connect(port)
            assert rude is not None, "server did not come up"
            rude.sendall(pose * 4)
            rude.setsockopt(socket.SOL_SOCKET, socket.SO_LINGER, b"\x01\x00\x00\x00\x00\x00\x00\x00")  # RST on close
            rude.close()
        time.sleep(0.5)
        assert srv.poll() is None, "the server died of a disconnected client"
        s = _connect(port)
        assert s is not None
        s.settimeout(60)
        quit_msg = np.zeros(16, np.float32)
        quit_msg[:1] = np.frombuffer(b"QUIT", np.float32)
        s.sendall(quit_msg.tobytes())              # without the hook: a pose like any other -> one frame comes back
        _recv_exact(s, 3 * W * H)
        assert srv.poll() is None
        # 80 views in one extended request = two chunks (the server renders up to 64 views per launch), answers in request order
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        views = [syn.orbit_pose(4.5 * i, 25.0) for i in range(80)]
        msg = b"NRF1" + np.uint32(len(views)).tobytes()
        for p in views:
            msg += cam.tobytes() + np.ascontiguousarray(p, np.float32).tobytes()
        s.sendall(msg)
        frames = [np.frombuffer(_recv_exact(s, 3 * W * H), np.uint8).reshape(H, W, 3) for _ in views]
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        for i in (0, 63, 64, 79):
            ctx.render(cam, views[i])
            np.testing.assert_array_equal(frames[i], ctx.read_u8()[0])
        ctx.close()
        s.close()
        assert srv.poll() is None
    finally:
        srv.kill()
        srv.wait(timeout=20)


@pytest.mark.gpu
def test_render_server_stalled_client_does_not_hold_the_gpu(snapshot):
    """a client that sends an extended request of many views and then reads nothing pinned its replies' host-frame
    slot for ever, and with it the worker of that GPU -- for every other client.  Now the images waiting behind a stalled
    socket are copied out of the slot when the worker wants it back (within a budget of spilled bytes), and a consumer that
    takes no byte for NRF_SERVER_SEND_TIMEOUT_S is dropped.  A stalled NRF1 client beside a normal one: the normal
    client's 40 sequential requests must all be answered (bytes equal to the binding's render) while the stalled one
    still holds its connection; the statistics then show evictions, and the stalled connection is closed by the server."""
    path, desc, keep, cfg = snapshot
    W, H, port = 640, 480, 23463  # 0.9 MB per image: a 24-view reply is far more than a socket buffer takes
    env = dict(SERVER_TEST_ENV, NRF_SERVER_BIND="127.0.0.1", NRF_SERVER_SEND_TIMEOUT_S="3")
    srv = subprocess.Popen([str(HOST / "render_server"), str(port), str(path), str(W), str(H)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=env)
    try:
        cam = np.array([840, 840, 339, 590], np.float32) * (np.float32(W) / np.float32(1080.0))
        stalled = _connect(port)
        assert stalled is not None, "server did not come up"
        stalled.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4096)
        views = [syn.orbit_pose(11.0 * i, 20.0) for i in range(24)]
        msg = b"NRF1" + np.uint32(len(views)).tobytes()
        for p in views:
            msg += cam.tobytes() + np.ascontiguousarray(p, np.float32).tobytes()
        stalled.sendall(msg)  # ... and never reads
        time.sleep(0.3)
        good = _connect(port)
        good.settimeout(20)  # a worker stuck behind the stalled client's slot would time this out
        poses = [syn.orbit_pose(9.0 * i, 30.0) for i in range(40)]
        frames = []
        t0 = time.time()
        for p in poses:
            good.sendall(np.ascontiguousarray(p, np.float32).tobytes())
            frames.append(np.frombuffer(_recv_exact(good, 3 * W * H), np.uint8).reshape(H, W, 3))
        served_s = time.time() - t0
        ctx = nh.NerfHip(0)
        ctx.load_model(desc)
        ctx.set_resolution(W, H)
        for i in (0, 1, 2, 20, 39):
            ctx.render(cam, poses[i])
            np.testing.assert_array_equal(frames[i], ctx.read_u8()[0])
        ctx.close()
        stat = np.zeros(16, np.float32)
        stat[:1] = np.frombuffer(b"STAT", np.float32)
        good.sendall(stat.tobytes())
        line = _recv_exact(good, 256).rstrip(b"\0").decode().split()
        fields = dict(zip(line[0::2], line[1::2]))
        assert int(fields["evictions"]) >= 1, fields  # the stalled client's images were copied out of a slot the worker wanted
        assert served_s < 15, served_s
        # the stalled client takes nothing: after the send timeout the server drops it
        time.sleep(4.5)
        good.sendall(stat.tobytes())
        fields = dict(zip(*[iter(_recv_exact(good, 256).rstrip(b"\0").decode().split())] * 2))
        assert int(fields["dropped_slow"]) >= 1, fields
        assert srv.poll() is None
        good.close()
        stalled.close()
    finally:
        srv.kill()
        srv.wait(timeout=20)


def test_snapshot_parser_rejects_hostile_input(tmp_path):
    """The msgpack reader trusts nothing: an array/map length larger than the bytes that follow, or nesting deep
    enough to exhaust the stack, is an error message -- not a 16 GiB reservation or a crash."""
    cases = {
        "huge_array": b"\x81\xa8snapshot\xdd\xff\xff\xff\xff\x01\x02",            # array32 of 2^32-1 elements, 2 present
        "huge_map": b"\x81\xa8snapshot\xdf\xff\xff\xff\xff",                        # map32 of 2^32-1 pairs
        "deep": b"\x81\xa8snapshot" + b"\x91" * 100000 + b"\x00",                      # 100 000 nested one-element arrays
        "zero_F": None,
    }
    import msgpack
    cases["zero_F"] = msgpack.packb({"encoding": {"otype": "HashGrid", "n_features_per_level": 0, "n_features": 32},
                                     "snapshot": {"aabb": [-1, -1, -1, 1, 1, 1], "density_grid_size": 1, "density_grid": [0.0],
                                                  "params": [0.0]}}, use_single_float=True)
    for name, blob in cases.items():
        f = tmp_path / f"{name}.msgpack"
        f.write_bytes(blob)
        r = _info(f)
        assert r.returncode == 1, (name, r.returncode, r.stderr[-300:])   # 1 = caught exception; a signal would be negative
        assert "error" in r.stderr.lower() or "msgpack" in r.stderr or "must be" in r.stderr, (name, r.stderr[-300:])


@pytest.mark.gpu
def test_snapshot_without_density_grid_renders_in_both_mirrors(tmp_path, snapshot):
    """f4 through the callers: a snapshot that carries no density grid (the reference's load_snapshot would throw on
    the missing key) is completed by NerfRender::generate_density_grid -- C++ testbed and Python mirror agree bit for
    bit, and with the HIP context's own generate + render."""
    path, desc, keep, cfg = snapshot
    bare = tmp_path / "nogrid.msgpack"
    syn.write_snapshot(bare, cfg, keep[0], None)
    W, H = 96, 64
    r = subprocess.run([str(HOST / "testbed"), str(bare), str(W), str(H), str(tmp_path) + "/"], capture_output=True, text=True,
                       timeout=180)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "density grid generated from the network" in r.stdout
    got = np.fromfile(tmp_path / "image.rgb", np.uint8).reshape(H, W, 3)
    s = np.float32(W) / np.float32(500.0)
    cam = np.array([3550.115 / 8, 3554.515 / 8, 3010.45 / 8, 1996.027 / 8], np.float32) * s
    render = nh.NerfRender(devices=[0])
    render.reload_network_from_file(bare)
    render.set_resolution((W, H))
    rgb, _ = render.render_frame(cam, syn.REFERENCE_MAIN_POSE)
    render.close()
    np.testing.assert_array_equal(got, rgb)
    d0, k0 = nh.desc_from_config(syn.read_snapshot(bare))
    ctx = nh.NerfHip(0)
    ctx.load_model(d0)
    ctx.generate_density_grid(16, 0.95)
    ctx.set_resolution(W, H)
    ctx.render(cam, syn.REFERENCE_MAIN_POSE)
    np.testing.assert_array_equal(ctx.read_u8()[0], got)
    ctx.close()
    assert got.min() < 250
