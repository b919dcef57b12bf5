"""MIOpen keeps the solver it found fastest for a convolution in a per-user find-db (~/.config/miopen/*.ufdb.txt) and reuses the entry in
every later process -- whatever solver set that process may use.  A run with torch.backends.cudnn.deterministic = True (MIOpen restricted to
its deterministic solvers) therefore leaves entries that later ordinary runs pick up: measured on MI355X, the 64 x 256^2 training step
went 60 ms -> 485 ms after one deterministic step on the same account, and back to 60 ms with a separate find-db.  `use_private_find_db`
points MIOpen at a directory of its own per purpose, before the first convolution of the process."""
import os


def use_private_find_db(tag: str, force: bool = False) -> str:
    """Give this process (and its children) the find-db directory ~/.config/miopen_<tag> unless MIOPEN_USER_DB_PATH is already set
    (`force` overrides).  Must run before MIOpen is first used.  Returns the directory in effect."""
    if force or not os.environ.get("MIOPEN_USER_DB_PATH"):
        path = os.path.join(os.path.expanduser("~"), ".config", "miopen_" + tag)
        try:
            os.makedirs(path, exist_ok=True)
        except OSError:
            import tempfile
            path = tempfile.mkdtemp(prefix="miopen_" + tag + "_")
        os.environ["MIOPEN_USER_DB_PATH"] = path
    return os.environ["MIOPEN_USER_DB_PATH"]
