"""Chunks in flight: the gort_pipe C ABI, the pinned / staged host entry points and the three-stage pipeline of
the `gortt` executable (parse | GPU | format).  Reference interface: the per-line loop gortt.c:232-329."""
import os
import subprocess

import numpy as np
import pytest

from conftest import relerr
from gort_amd import api
from oracle import oracle as O

pytestmark = pytest.mark.gpu
REGRESSION = 1e-9


def _lines(rng, n, distinct=True):
    sza = rng.uniform(0, 89, n) if distinct else rng.integers(0, 90, n).astype(float)
    return np.stack([rng.uniform(-89, 89, n), rng.uniform(0, 360, n), sza, rng.uniform(0, 360, n)], 1)


@pytest.fixture(scope="module")
def eng():
    e = api.Engine()
    e.set_canopy(api.gap_probabilities(api.make_canopy(lai=4.0)))
    yield e
    e.close()


def test_pipe_chunks_in_flight_equal_one_call(eng):
    """10 chunks through a 3-slot pipe (producer runs ahead of the consumer) == the same lines in one call."""
    rng = np.random.default_rng(11)
    wl = np.linspace(400.0, 2500.0, 700)
    eng.set_spectra(*api.spectra(wl))
    ang = _lines(rng, 10 * 3000 - 1234, distinct=False)
    whole, sc_whole, K_whole = eng.rsurf_stream(ang, want_scomp=True)
    e_whole = eng.energy_stream(ang[:500])
    pipe = api.Pipe(eng, 3000, depth=3, flags=api.PIPE_SCOMP)
    got, got_sc, got_K = [], [], []
    sent = collected = 0
    k = 0
    nchunks = (ang.shape[0] + 2999) // 3000
    while collected < nchunks:
        while sent < nchunks and sent - collected < 3:
            a = pipe.acquire()
            n = min(3000, ang.shape[0] - sent * 3000)
            a[:n] = ang[sent * 3000: sent * 3000 + n]
            pipe.submit(n)
            sent += 1
        c = pipe.wait()
        assert np.array_equal(c["angles"], ang[collected * 3000: collected * 3000 + c["n"]])
        got.append(c["rsurf"].copy()); got_sc.append(c["scomp"].copy()); got_K.append(c["K"].copy())
        pipe.release()
        collected += 1
    pipe.close()
    assert np.array_equal(np.concatenate(got).view(np.int64), whole.view(np.int64))
    assert np.array_equal(np.concatenate(got_sc).view(np.int64), sc_whole.view(np.int64))
    assert np.array_equal(np.concatenate(got_K).view(np.int64), K_whole.view(np.int64))
    # energy through a pipe
    pipe = api.Pipe(eng, 500, depth=1, flags=api.PIPE_ENERGY_ONLY)
    a = pipe.acquire(); a[:500] = ang[:500]; pipe.submit(500)
    c = pipe.wait()
    assert c["rsurf"] is None and np.array_equal(c["energy"].view(np.int64), e_whole.view(np.int64))
    pipe.release(); pipe.close()
    # an empty chunk is a chunk
    pipe = api.Pipe(eng, 16, depth=2)
    pipe.acquire(); pipe.submit(0)
    assert pipe.wait()["n"] == 0
    pipe.release()
    # a chunk that failed when it was submitted is a chunk too: the consumer meets the error - with its cause -
    # in wait(), the slot is drained and released, and the pipe goes on working
    pipe.acquire()
    with pytest.raises(api.GortError, match="17 lines in a slot of 16"):
        pipe.submit(17)
    with pytest.raises(api.GortError, match="failed when it was submitted.*17 lines in a slot of 16"):
        pipe.wait()
    pipe.release()
    a = pipe.acquire(); a[:3] = ang[:3]; pipe.submit(3)
    c = pipe.wait()
    assert c["n"] == 3 and np.array_equal(c["rsurf"].view(np.int64), whole[:3].view(np.int64))
    pipe.release(); pipe.close()


def test_host_entry_pinned_equals_staged_equals_oracle(eng):
    """gort_rsurf_stream into pageable memory (pinned staging, 3 chunks in flight, threaded copy-out) and into a
    pinned buffer (one DMA) give the same bits; a sample of lines against the oracle."""
    rng = np.random.default_rng(12)
    wl = np.arange(400.0, 2501.0)
    rs, rl, tl = api.spectra(wl)
    eng.set_spectra(rs, rl, tl)
    ang = _lines(rng, 30011)
    staged, _, K = eng.rsurf_stream(ang)
    pin = api.PinnedArray((ang.shape[0], wl.size))
    pinned, _, K2 = eng.rsurf_stream(ang, out=pin.array)
    assert pinned is pin.array
    assert np.array_equal(staged.view(np.int64), pinned.view(np.int64)) and np.array_equal(K, K2)
    idx = np.sort(rng.choice(ang.shape[0], 30, replace=False))
    c = O.make_canopy(lai=4.0)
    ref, _, _ = O.rsurf_stream(c, ang[idx], *O.spectra(wl), want_K=False)
    assert relerr(staged[idx], ref, floor=1e-12) <= REGRESSION
    pin.free()
    # the energy entry point staged (3 x 2101 doubles per line)
    e_big = eng.energy_stream(ang[:4000])
    e_small = np.concatenate([eng.energy_stream(ang[i:i + 50]) for i in range(0, 200, 50)])
    assert np.array_equal(e_big[:200].view(np.int64), e_small.view(np.int64))


def test_pipe_indexed_energy_equals_dense_energy(eng):
    """GORT_PIPE_ENERGY_INDEXED: a chunk's `energy` holds its distinct rows and energy_index[line] picks the line's; chunk
    by chunk the same bits as the dense pipe, with and without the reflectances beside them, an empty chunk, a chunk of one
    line, and chunks in which every line has its own sun."""
    rng = np.random.default_rng(21)
    wl = np.linspace(400.0, 2500.0, 333)
    eng.set_spectra(*api.spectra(wl))
    ang = _lines(rng, 5 * 700, distinct=False)
    ang[700:1400, 2] = rng.uniform(0, 89, 700)                     # second chunk: nothing shared
    ang[:, 3] = rng.choice([0.0, 123.0], ang.shape[0])
    sizes = [700, 700, 0, 1, 700, 699]
    for flags in (api.PIPE_ENERGY, api.PIPE_ENERGY_ONLY):
        dense_pipe = api.Pipe(eng, 700, depth=2, flags=flags)
        pipe = api.Pipe(eng, 700, depth=3, flags=flags | api.PIPE_ENERGY_INDEXED)
        at = 0
        for n in sizes:
            for q in (dense_pipe, pipe):
                a = q.acquire(); a[:n] = ang[at:at + n]; q.submit(n)
            d, c = dense_pipe.wait(), pipe.wait()
            assert c["n"] == n == d["n"]
            if n == 0:
                assert c["energy_rows"] == 0 and c["energy"] is None
            else:
                assert d["energy_index"] is None and d["energy_rows"] == n
                assert 1 <= c["energy_rows"] <= n and c["energy"].shape == (c["energy_rows"], wl.size, 3)
                assert c["energy_rows"] == (n if at == 700 else len({(x, y) for x, y in ang[at:at + n, 2:].tolist()}))
                assert np.array_equal(c["energy"][c["energy_index"]].view(np.int64), d["energy"].view(np.int64))
                if flags == api.PIPE_ENERGY:
                    assert np.array_equal(c["rsurf"].view(np.int64), d["rsurf"].view(np.int64)) and np.array_equal(c["K"], d["K"])
                else:
                    assert c["rsurf"] is None
            dense_pipe.release(); pipe.release()
            at += n
        dense_pipe.close(); pipe.close()
    with pytest.raises(api.GortError, match="ENERGY_INDEXED without"):
        api.Pipe(eng, 16, depth=1, flags=api.PIPE_ENERGY_INDEXED)


def _suns_chunk(rng, n, distinct):
    a = _lines(rng, n, distinct=distinct)
    if not distinct:
        a[:, 3] = rng.choice([0.0, 123.0], n)
    return a


def test_pipe_indexed_energy_grows_a_slot(eng):
    """A slot of the indexed pipe starts with 16 MiB of row buffers (332 rows of 2101 bands) and exchanges them for bigger ones
    when a chunk has more distinct sun directions than that (gort_pipe.hip, gort_pipe_submit): 700 lines of 2101 bands with every
    line its own sun - what `gortt -energy` meets on a full-spectrum stream whose lines have their own suns, gortt.c:321-327 - then
    a chunk of few suns on the grown slot, growth of the other slot by a chunk in between the two capacities, and a chunk that
    fits what a slot grew to; chunk by chunk the bits of the dense pipe."""
    rng = np.random.default_rng(23)
    wl = np.arange(400.0, 2501.0)
    eng.set_spectra(*api.spectra(wl))
    assert (16 << 20) // (8 * 3 * wl.size) == 332
    plan = [(700, False), (700, True), (700, True), (600, False), (700, False), (500, True), (690, True), (0, False), (400, True)]
    dense_pipe = api.Pipe(eng, 700, depth=2, flags=api.PIPE_ENERGY_ONLY)
    pipe = api.Pipe(eng, 700, depth=2, flags=api.PIPE_ENERGY_ONLY | api.PIPE_ENERGY_INDEXED)
    for k, (n, distinct) in enumerate(plan):
        ang = _suns_chunk(rng, n, distinct)
        for q in (dense_pipe, pipe):
            a = q.acquire(); a[:n] = ang; q.submit(n)
        d, c = dense_pipe.wait(), pipe.wait()
        assert c["n"] == n == d["n"]
        if n:
            want_rows = n if distinct else len({(x, y) for x, y in ang[:, 2:].tolist()})
            assert c["energy_rows"] == want_rows and c["energy"].shape == (want_rows, wl.size, 3), k
            assert np.array_equal(c["energy"][c["energy_index"]].view(np.int64), d["energy"].view(np.int64)), k
        dense_pipe.release(); pipe.release()
    dense_pipe.close(); pipe.close()


@pytest.mark.ab
def test_pipe_slot_that_cannot_grow_fails_the_chunk_and_stays_whole(eng, monkeypatch):
    """The allocation of the bigger row buffers fails (GORT_PIPE_FAIL_GROW, measuring build): the chunk fails with GORT_ENOMEM at
    submit and at wait, the slot keeps the buffers it had - the chunk in flight beside it and the next chunk on the same slot are
    served - and the pipe closes cleanly."""
    rng = np.random.default_rng(24)
    wl = np.arange(400.0, 2501.0)
    eng.set_spectra(*api.spectra(wl))
    dense = api.Pipe(eng, 700, depth=1, flags=api.PIPE_ENERGY_ONLY)
    pipe = api.Pipe(eng, 700, depth=2, flags=api.PIPE_ENERGY_ONLY | api.PIPE_ENERGY_INDEXED)
    few, many, few2 = _suns_chunk(rng, 700, False), _suns_chunk(rng, 700, True), _suns_chunk(rng, 650, False)

    def dense_rows(ang):
        a = dense.acquire(); a[:len(ang)] = ang; dense.submit(len(ang))
        r = dense.wait()["energy"].copy(); dense.release()
        return r
    a = pipe.acquire(); a[:700] = few; pipe.submit(700)              # slot 0, in flight
    monkeypatch.setenv("GORT_PIPE_FAIL_GROW", "1")
    a = pipe.acquire(); a[:700] = many
    with pytest.raises(api.GortError, match="distinct sun directions"):
        pipe.submit(700)                                              # slot 1 cannot grow
    monkeypatch.delenv("GORT_PIPE_FAIL_GROW")
    c = pipe.wait()
    assert np.array_equal(c["energy"][c["energy_index"]].view(np.int64), dense_rows(few).view(np.int64))
    pipe.release()
    with pytest.raises(api.GortError):
        pipe.wait()                                                   # the failed chunk reports its failure
    pipe.release()
    a = pipe.acquire(); a[:650] = few2; pipe.submit(650)             # slot 0 again
    a = pipe.acquire(); a[:700] = many; pipe.submit(700)             # slot 1 again: grows now
    c = pipe.wait()
    assert np.array_equal(c["energy"][c["energy_index"]].view(np.int64), dense_rows(few2).view(np.int64))
    pipe.release()
    c = pipe.wait()
    assert c["energy_rows"] == 700 and np.array_equal(c["energy"][c["energy_index"]].view(np.int64), dense_rows(many).view(np.int64))
    pipe.release()
    pipe.close(); dense.close()


def test_cli_energy_full_spectrum_stream_of_own_suns_grows_the_slots():
    """`gortt -energy --binary-out` on 1500 lines x 2101 bands, every line its own sun direction (one chunk, 1500 distinct rows: the
    slot's row buffers grow from 332 rows), and the same stream with three sun directions: the bytes of GORTT_ENERGY_DENSE=1."""
    rng = np.random.default_rng(25)
    wl = np.arange(400, 2501)
    head = ("1500 %d %s\n" % (len(wl), " ".join("%d" % w for w in wl))).encode()
    for distinct in (True, False):
        ang = np.round(_suns_chunk(rng, 1500, distinct), 4)
        text = head + "".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in ang).encode()
        rc, out, err = _gortt(["-LAI", "4.0", "-energy", "--binary-out"], text)
        rc2, out2, err2 = _gortt(["-LAI", "4.0", "-energy", "--binary-out"], text, {"GORTT_ENERGY_DENSE": "1"})
        assert rc == 0 and rc2 == 0 and err == b"" and err2 == b"", (err, err2)
        assert len(out) > 1500 * 2101 * 8 * 4 and out == out2, distinct


def test_cli_energy_rows_formatted_once_equal_rows_formatted_per_line():
    """`gortt -energy`: the indexed pipe (each distinct albedo row copied and formatted once per chunk) writes the bytes of
    the dense path (GORTT_ENERGY_DENSE=1: a row per line, as gortt.c:321-327 evaluates them) - text over several chunks and
    formatting threads, binary, with -prnprop and -prnspec beside it, every line its own sun, and an empty stream."""
    rng = np.random.default_rng(22)
    n, wl = 6000, np.linspace(400, 2500, 211).round(1)
    ang = np.round(_lines(rng, n, distinct=False), 3)
    ang[::7, 2] *= -1.0
    ang[:, 3] = rng.choice([0.0, 45.0, 200.0], n)
    head = ("%d %d %s\n" % (n, len(wl), " ".join("%g" % w for w in wl))).encode()
    text = head + "".join("%.3f %.3f %.3f %.3f\n" % tuple(r) for r in ang).encode()
    small = {"GORTT_CHUNK_MB": "1"}
    for args in (["-energy"], ["-energy", "-prnprop"], ["-energy", "-prnspec"], ["-energy", "--binary-out"]):
        rc, out, err = _gortt(["-LAI", "4.0"] + args, text, small)
        rc2, out2, err2 = _gortt(["-LAI", "4.0"] + args, text, dict(small, GORTT_ENERGY_DENSE="1"))
        assert rc == 0 and rc2 == 0 and err == b"" and err2 == b"", (args, err, err2)
        assert out == out2, args
        rc3, out3, _ = _gortt(["-LAI", "4.0"] + args, text, dict(small, GORTT_THREADS="3"))
        assert rc3 == 0 and out3 == out
    # every line its own sun direction, one big chunk
    ang2 = np.round(_lines(rng, 1500), 4)
    text2 = ("%d 5 450 550 650 850 1600\n" % 1500).encode() + "".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in ang2).encode()
    rc, out, err = _gortt(["-LAI", "2.5", "-energy"], text2)
    rc2, out2, err2 = _gortt(["-LAI", "2.5", "-energy"], text2, {"GORTT_ENERGY_DENSE": "1"})
    assert rc == 0 and rc2 == 0 and out == out2 and out.count(b"\n") == 1501
    # no lines at all; and no wavelengths (`N 0`: -energy prints nothing per line, gortt.c:323)
    for stdin in (b"0 2 500 600\n", b"2 0\n10 0 30 0\n20 0 40 0\n"):
        rc, out, err = _gortt(["-LAI", "4.0", "-energy"], stdin)
        rc2, out2, err2 = _gortt(["-LAI", "4.0", "-energy"], stdin, {"GORTT_ENERGY_DENSE": "1"})
        assert (rc, out, err) == (rc2, out2, err2) and rc == 0


def _gortt(args, stdin_bytes, env=None):
    e = dict(os.environ)
    e.update(env or {})
    run = subprocess.run([api.GORTT_BIN] + args, input=stdin_bytes, capture_output=True, timeout=600, env=e)
    return run.returncode, run.stdout, run.stderr


def test_cli_pipeline_many_chunks_in_order_and_devices():
    """A text stream of several chunks (chunk size follows the band count): rows leave in input order, and
    --gpus 1, the default and two pipes on one device (GORTT_DEVICES=0,0: the multi-device path of a box with one
    GPU) write identical bytes; binary output carries the same numbers."""
    rng = np.random.default_rng(13)
    n, wl = 9000, np.linspace(400, 2500, 1500).round(2)
    ang = np.round(_lines(rng, n, distinct=False), 4)
    head = ("%d %d %s\n" % (n, len(wl), " ".join("%g" % w for w in wl))).encode()
    text = head + "".join("%.4f %.4f %.4f %.4f\n" % tuple(r) for r in ang).encode()
    rc, out, err = _gortt(["-LAI", "4.0", "-prnprop"], text)
    assert rc == 0 and err == b""
    lines = out.split(b"\n")
    assert lines[0] + b"\n" == head and len(lines) == n + 2
    first = np.array([[float(t) for t in ln.split()[:4]] for ln in lines[1:n + 1]])
    assert np.allclose(first, ang, atol=5.1e-7)                      # order kept
    for env, extra in (({}, ["--gpus", "1"]), ({"GORTT_DEVICES": "0,0"}, []), ({"GORTT_THREADS": "1"}, [])):
        rc2, out2, err2 = _gortt(["-LAI", "4.0", "-prnprop"] + extra, text, env)
        assert rc2 == 0 and out2 == out, (env, extra)
    rc, bout, err = _gortt(["-LAI", "4.0", "-prnprop", "--binary-out"], text, {"GORTT_DEVICES": "0,0"})
    assert rc == 0
    rows = np.frombuffer(bout[len(head):], dtype="<f8").reshape(n, 4 + len(wl) + 4)
    assert np.array_equal(rows[:, :4], ang)
    vals = np.array([[float(t) for t in ln.replace(b"[", b" ").replace(b"]", b" ").split()] for ln in lines[1:n + 1:97]])
    assert np.allclose(vals, rows[::97], rtol=0, atol=5.0001e-7, equal_nan=True)
    # a bad line in the third chunk: every row in front of it is written, then the reference's message
    bad = text.split(b"\n")
    bad[7000] = b"12 abc"
    rc, out3, err3 = _gortt(["-LAI", "4.0", "-prnprop"], b"\n".join(bad))
    assert rc != 0 and b"error on input, line 7000" in err3
    assert out3 == b"\n".join(lines[:7000]) + b"\n"
    # fewer lines than announced: all rows, then the count message (gortt.c:331-336)
    rc, out4, err4 = _gortt(["-LAI", "4.0", "-prnprop"], b"\n".join(text.split(b"\n")[:5001]) + b"\n")
    assert rc != 0 and out4 == b"\n".join(lines[:5001]) + b"\n"
    assert b"expected number of angles (9000) does not match with number found (5000)" in err4


def test_cli_bulk_reader_edge_cases():
    """The text reader takes stdin in 8 MB blocks and cuts lines with memchr: lines of very different lengths across
    block boundaries, CRLF endings, a last line without a newline, a blank line (an error at the right line number, as
    the reference's sscanf would report it: gortt.c:234-237), extra columns."""
    rng = np.random.default_rng(17)
    n, wl = 140000, [650.0, 865.0]
    ang = np.round(_lines(rng, n, distinct=False), 3)
    head = ("%d %d 650 865\n" % (n, len(wl))).encode()
    def line(i, r):
        pad = " " * int(rng.integers(0, 40)) if i % 7 == 0 else ""
        tail = (" extra %d columns" % i) * int(rng.integers(0, 6)) if i % 11 == 0 else ""
        eol = "\r\n" if i % 5 == 0 else "\n"
        return ("%s%.3f %.3f\t%.3f   %.3f%s%s" % (pad, r[0], r[1], r[2], r[3], tail, eol)).encode()
    body = b"".join(line(i, r) for i, r in enumerate(ang))
    assert len(body) > (8 << 20) // 2                                   # several chunks, more than one block with the rows below
    text = ("%d %d 650 865\n" % (2 * n, len(wl))).encode() + body + body[:-1]      # ... and no newline at the very end
    rc, out, err = _gortt(["-LAI", "4.0", "--binary-out"], text)
    assert rc == 0 and err == b"", err[-300:]
    rows = np.frombuffer(out[len(text.split(b"\n", 1)[0]) + 1:], dtype="<f8").reshape(2 * n, 4 + len(wl))
    assert np.array_equal(rows[:n, :4], ang) and np.array_equal(rows[n:, :4], ang)
    assert np.array_equal(rows[:n, 4:].view(np.int64), rows[n:, 4:].view(np.int64))
    c = O.make_canopy(lai=4.0)
    idx = np.sort(rng.choice(n, 25, replace=False))
    ref, _, _ = O.rsurf_stream(c, ang[idx], *O.spectra(wl), want_K=False)
    assert relerr(rows[idx, 4:], ref, floor=1e-12) <= REGRESSION
    # a blank line in the second block: rows in front of it are written, the message names its line
    parts = text.split(b"\n")
    k = 200001
    parts[k] = b""
    rc, out2, err2 = _gortt(["-LAI", "4.0", "--binary-out"], b"\n".join(parts))
    assert rc != 0 and ("error on input, line %d" % k).encode() in err2
    assert len(out2) == len(parts[0]) + 1 + (k - 1) * 8 * (4 + len(wl))


def _tables(c):
    return np.array(list(c.p_n0) + list(c.epgap) + [c.k_open, c.k_openep]).view(np.int64)


def test_gap_tables_cached_per_crown_geometry_in_process_and_on_disk(tmp_path):
    """SURVEY.md 8(f) row 3.  In process: a member whose crown geometry has been seen takes its tables from the cache
    (same bits), only the others go to the device, in mixed batches too.  `gortt --lut-cache DIR`: the second run
    reads what the first one kept, writes the same bytes, and the file holds exactly what the device computed."""
    api.gap_cache_clear()
    a = api.gap_probabilities(api.make_canopy(lai=4.0))
    assert api.gap_cache_stats() == (0, 1, 1)
    b = api.gap_probabilities(api.make_canopy(lai=4.0, beta=0.25))       # -beta is not part of the geometry
    assert api.gap_cache_stats() == (1, 1, 1)
    assert np.array_equal(_tables(a), _tables(b))
    batch = [api.make_canopy(lai=2.5), api.make_canopy(lai=4.0), api.make_canopy(newstyle=(2.0, 2.0, 0.6), lai=3.3),
             api.make_canopy(lai=4.0, q08=True)]
    api.gap_probabilities(batch)
    assert api.gap_cache_stats() == (2, 4, 4)                            # one hit, three computed
    api.gap_cache_clear()
    assert api.gap_cache_stats() == (0, 0, 0)
    for m, kw in zip(batch, (dict(lai=2.5), dict(lai=4.0), dict(newstyle=(2.0, 2.0, 0.6), lai=3.3), dict(lai=4.0, q08=True))):
        alone = api.gap_probabilities(api.make_canopy(**kw))             # each computed on its own, cache empty
        assert np.array_equal(_tables(m), _tables(alone)), kw
    assert not np.array_equal(_tables(batch[1]), _tables(batch[3]))     # q08 is part of the key
    # the CLI
    text = b"3 4 450 600 800 1000\n10 0 30 20\n-40 10 55 200\n88 0 89 180\n"
    d = str(tmp_path)
    rc0, plain, _ = _gortt(["-LAI", "4.0", "-prnprop"], text)
    rc1, first, err1 = _gortt(["-LAI", "4.0", "-prnprop", "--lut-cache", d], text, {"GORTT_VERBOSE": "1"})
    rc2, second, err2 = _gortt(["-LAI", "4.0", "-prnprop", "--lut-cache", d], text, {"GORTT_VERBOSE": "1"})
    assert rc0 == rc1 == rc2 == 0 and plain == first == second
    assert b"gap tables computed, kept in" in err1 and b"gap tables from" in err2
    kept = api.make_canopy(lai=4.0)
    assert api.lut_cache_load(d, kept) and np.array_equal(_tables(kept), _tables(a))
    # another stand: its own entry; an unwritable cache directory is a warning, not a failure
    rc3, _, err3 = _gortt(["-LAI", "2.0", "--lut-cache", d], text, {"GORTT_VERBOSE": "1"})
    assert rc3 == 0 and b"computed" in err3 and len(os.listdir(d)) == 2
    rc4, out4, err4 = _gortt(["-LAI", "4.0", "-prnprop", "--lut-cache", os.path.join(d, "missing")], text)
    assert rc4 == 0 and out4 == plain and b"warning" in err4
