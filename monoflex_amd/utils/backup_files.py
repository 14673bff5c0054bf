"""`sync_root(root, target)`: copy the source tree of a run (python / yaml / HIP sources) next to its logs
(reference utils/backup_files.py:56-75; called by tools/plain_train_net.py before training)."""
import os
import shutil

KEEP = (".py", ".yaml", ".hip", ".h", ".md")
SKIP_DIRS = {".git", "__pycache__", "gpurun_out", "build", "golden", "profiles"}


def sync_root(root, target):
    n = 0
    for d, dirs, files in os.walk(root):
        dirs[:] = [x for x in dirs if x not in SKIP_DIRS and not os.path.abspath(os.path.join(d, x)).startswith(os.path.abspath(target))]
        for f in files:
            if f.endswith(KEEP):
                dst = os.path.join(target, os.path.relpath(os.path.join(d, f), root))
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copy2(os.path.join(d, f), dst)
                n += 1
    return n
