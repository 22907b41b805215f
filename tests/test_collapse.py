"""mtsv-collapse: the reference's own test vectors (src/collapse.rs:788-817)
and the chunk-mode equivalence needed for BASELINE config 5."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mtsv_tools_amd", "bin", "mtsv-collapse")


def collapse(tmp_path, texts, *extra):
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / f"in{i}.txt"
        p.write_text(t)
        paths.append(str(p))
    out = tmp_path / "out.txt"
    r = subprocess.run([BIN, "-o", str(out), *extra, *paths], capture_output=True, text=True)
    return r, (out.read_text() if out.exists() else "")


def test_min_edit_per_taxid(tmp_path):
    # collapse.rs:788-803
    r, text = collapse(tmp_path, ["r1:1=5,2=9\nr2:3=4", "r1:1=2,2=10\nr2:3=1"])
    assert r.returncode == 0 and text == "r1:1=2,2=9\nr2:3=1\n"
    r2, text2 = collapse(tmp_path, ["r1:1=2,2=10\nr2:3=1", "r1:1=5,2=9\nr2:3=4"])
    assert text2 == text  # independent of file order


def test_min_edit_per_taxid_gi_with_offset_tiebreak(tmp_path):
    # collapse.rs:805-817
    r, text = collapse(tmp_path, ["r1:1-5-3=7,1-5-2=4\nr2:2-9-1=3", "r1:1-5-4=5,2-8-1=6\nr2:2-9-1=2"], "--mode", "taxid-gi")
    assert r.returncode == 0 and text == "r1:1-5-2=4,2-8-1=6\nr2:2-9-1=2\n"


def test_errors_and_report(tmp_path):
    r, _ = collapse(tmp_path, ["r1:1=5"], "--mode", "taxid-gi")       # Missing GI for taxid-gi collapse
    assert r.returncode == 101
    r, _ = collapse(tmp_path, ["r1:1-2-3=5,1-2=4"], "--mode", "taxid-gi")  # mixed offset formats
    assert r.returncode == 101
    r, _ = collapse(tmp_path, ["no colon"])
    assert r.returncode == 101
    rep = tmp_path / "rep.tsv"
    r, text = collapse(tmp_path, ["a:1=0,2=3\nb:1=2\nc:1=1,2=1"], "--report", str(rep))
    assert r.returncode == 0
    rows = [l.split("\t") for l in rep.read_text().splitlines()]
    assert rows[0][0] == "taxid" and rows[1][:2] == ["1", "1"] and rows[1][3] == "1" and rows[1][5] == "1"
    assert rows[2][0] == "2" and rows[2][7] == "1" and rows[2][5] == "1"


def test_external_sort_runs_give_the_in_memory_result(tmp_path):
    """collapse.rs:427-475,665: inputs are sorted in bounded runs spilled to a temporary directory and merged;
    tiny runs (64 bytes instead of 128 MiB) must not change a byte of the output, and leave no files behind."""
    import random
    rng = random.Random(5)
    files = []
    for f in range(4):
        lines = []
        for r in rng.sample(range(600), 400):
            hits = ",".join(f"{rng.randrange(1, 40)}-{rng.randrange(1, 5)}-{rng.randrange(0, 9000)}={rng.randrange(0, 20)}"
                            for _ in range(rng.randrange(1, 6)))
            lines.append(f"read_{r}:x:{hits}" if r % 7 == 0 else f"read_{r}:{hits}")  # ids may hold ':' (rsplit)
        files.append("\n".join(lines) + "\n")
    tmp = tmp_path / "tmpdir"
    tmp.mkdir()
    for mode in ("taxid", "taxid-gi"):
        r0, want = collapse(tmp_path, files, "--mode", mode)
        assert r0.returncode == 0 and want.count("\n") == len({l.rsplit(":", 1)[0] for f in files for l in f.splitlines()})
        paths = [str(tmp_path / f"in{i}.txt") for i in range(4)]
        out = tmp_path / "spilled.txt"
        env = dict(os.environ, MTSV_COLLAPSE_CHUNK_BYTES="64", TMPDIR=str(tmp))
        r1 = subprocess.run([BIN, "-o", str(out), "--mode", mode, *paths], capture_output=True, text=True, env=env)
        assert r1.returncode == 0, r1.stderr
        assert out.read_text() == want
        assert os.listdir(tmp) == []
    ids = [l.rsplit(":", 1)[0] for l in want.splitlines()]
    assert ids == sorted(ids)
