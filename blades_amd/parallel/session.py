"""``get_session(client_id)`` — the handle client code uses to reach its worker's
shared Task (reference ``fllib/core/execution/session.py:6-12``)."""
from .worker import current_worker


def get_session(client_id):
    w = current_worker()
    if w is None:
        raise RuntimeError("No worker in this process; build a WorkerGroup (or Algorithm) first.")
    w.switch_client(client_id)
    return w
