"""profiles/latest_traffic.json -- the file bench.py quotes `roofline.traffic` and `roofline.valu` from -- must describe the
kernels that ship: its git_revision has to be an ancestor of (or equal to) HEAD, and no kernel source may have changed since -- or, if
one has, the in-tree library's machine code must still be what was profiled (the file's code_hashes, or at least the profiled
kernel's own instructions: kernel_hash; tools/code_hash.py).
(Where there is no git history -- the GPU box gets a snapshot without .git -- the test has nothing to check and is skipped.)"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_FILES = ["clsim_amd/csrc/prop_kernel.hip", "clsim_amd/csrc/prop_pool_kernel.hip", "clsim_amd/csrc/prop_tab_kernel.hip",
                "clsim_amd/csrc/prop_keep_kernel.hip", "clsim_amd/csrc/prop_pool_keep_kernel.hip",
                "clsim_amd/csrc/prop_device.hip.h", "clsim_amd/csrc/detmath.hip.h", "clsim_amd/csrc/kparams.h", "clsim_amd/csrc/Makefile"]


def git(*args):
    return subprocess.run(["git", "-C", ROOT] + list(args), capture_output=True, text=True)


def test_traffic_profile_was_taken_at_the_shipped_kernels():
    if not os.path.isdir(os.path.join(ROOT, ".git")) or git("rev-parse", "HEAD").returncode != 0:
        pytest.skip("no git history here")
    with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
        prof = json.load(f)
    rev = prof["git_revision"]
    assert git("cat-file", "-e", rev + "^{commit}").returncode == 0, "profiles/latest_traffic.json names an unknown revision: " + rev
    assert git("merge-base", "--is-ancestor", rev, "HEAD").returncode == 0, rev + " is not an ancestor of HEAD"
    changed = git("diff", "--name-only", rev, "HEAD", "--", *KERNEL_FILES).stdout.split()
    # uncommitted edits count as well
    changed += git("diff", "--name-only", "HEAD", "--", *KERNEL_FILES).stdout.split()
    if changed and prof.get("code_hashes"):
        # the sources moved on, but did the machine code?  (comment edits, renames: tools/code_hash.py of the in-tree library)
        out = subprocess.run(["python3", os.path.join(ROOT, "tools", "code_hash.py")], capture_output=True, text=True)
        now = [line.split()[0] for line in out.stdout.splitlines() if line.strip()]
        if out.returncode == 0 and now == prof["code_hashes"]:
            changed = []
    if changed and prof.get("kernel_hash"):
        # other kernels of the library changed (the generic instantiations got a medium kind, say): is the PROFILED kernel still the
        # same instructions?  (tools/code_hash.py --kernels: per kernel symbol, addresses and branch targets taken out)
        out = subprocess.run(["python3", os.path.join(ROOT, "tools", "code_hash.py"), "--kernels"], capture_output=True, text=True)
        want = prof["kernel_hash"]
        for line in out.stdout.splitlines():
            fields = line.split()
            if len(fields) == 3 and fields[2] == want["symbol"] and fields[0] == want["sha16"] and int(fields[1]) == want["instructions"]:
                changed = []
    assert not changed, "kernel sources changed since the profile of %s was taken: %s -- rerun tools/profile_round.sh + tools/make_latest_traffic.py" % (rev, sorted(set(changed)))
    assert "prop_pool_kernel<1, true, false, false, true, false>" in prof["kernel"]        # (MED, TILT, ANISO, FLASHER, FAST, KEEP)
