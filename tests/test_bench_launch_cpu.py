"""CPU: bench.py's own launcher -- `python bench.py --gpus N` without torch.distributed.run starts N ranks with the rendezvous
environment, and a rank count that does not match --gpus is refused (the multi-GPU line can never silently be a 1-GPU run)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
                      "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n")
    assert bench.spawn_ranks(3, argv=[str(tmp_path)], script=str(script)) == 0
    seen = [open(tmp_path / ("rank%d" % r)).read().split() for r in range(3)]
    assert [s[0] for s in seen] == ["0", "1", "2"] and [s[1] for s in seen] == ["0", "1", "2"]
    assert all(s[2] == "3" and s[3] == "127.0.0.1" for s in seen)
    assert len({s[4] for s in seen}) == 1 and int(seen[0][4]) > 0


def test_rank_count_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 2 and b"WORLD_SIZE=1" in r.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 2


def test_reference_cli_timing_lines_parse():
    import re
    err = ("falcon_print_timings: batch eval time =    12.34 ms /   128 tokens (    0.10 ms per token, 10372.77 tokens per second)\n"
           "falcon_print_timings:       eval time =   140.00 ms /   127 runs   (    1.10 ms per token,   907.14 tokens per second)\n")
    src = open(os.path.join(ROOT, "bench.py")).read()
    pats = re.findall(r're\.search\(r"(.*?)", err\)', src)
    assert len(pats) == 2
    a, b = (re.search(p, err) for p in pats)
    assert a and float(a.group(4)) == 10372.77 and int(a.group(2)) == 128
    assert b and float(b.group(4)) == 907.14 and int(b.group(2)) == 127
