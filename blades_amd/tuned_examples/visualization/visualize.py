"""Accuracy-vs-#malicious-clients plot, one panel per aggregator, one line per server momentum — the plot
of the reference's ``tuned_examples/visualization/visualize.py:8-49`` drawn from the same result-file
contract: ``<results>/<experiment>/<trial>/params.pkl`` (``server_config.aggregator.type``,
``server_config.optimizer.momentum``, ``num_malicious_clients``) and the last line of ``result.json``
(``acc_top_1``).  matplotlib only (seaborn is not a dependency here); several trials with the same key
(different seeds) are drawn as mean with a min–max band.

usage: python -m blades_amd.tuned_examples.visualization.visualize [--log-dir ~/blades_results/fedavg_blades]
                                                                    [--metric acc_top_1] [--out plot.png]
"""
import argparse
import json
import os
import pickle
from collections import defaultdict


def collect(log_dir: str, metric: str = "acc_top_1"):
    """[{Aggregator, Momentum, num_malicious_clients, Accuracy}] — one item per finished trial."""
    rows = []
    log_dir = os.path.expanduser(log_dir)
    for sub in sorted(os.listdir(log_dir)):
        d = os.path.join(log_dir, sub)
        pf, rf = os.path.join(d, "params.pkl"), os.path.join(d, "result.json")
        if not (os.path.isdir(d) and os.path.exists(pf) and os.path.exists(rf)):
            continue
        with open(pf, "rb") as f:
            params = pickle.load(f)  # written by this framework's own trial runner
        last = None
        with open(rf) as f:
            for line in f:
                if line.strip():
                    rec = json.loads(line)
                    if metric in rec:
                        last = rec
        if last is None:
            continue
        agg = params["server_config"]["aggregator"]
        rows.append({
            "Aggregator": agg["type"] if isinstance(agg, dict) else str(agg),
            "Momentum": params["server_config"].get("optimizer", {}).get("momentum", 0.0),
            "num_malicious_clients": params.get("num_malicious_clients", 0),
            "Accuracy": last[metric],
        })
    return rows


def plot(rows, out=None, show=False):
    import matplotlib

    if not show:
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    aggs = sorted({r["Aggregator"] for r in rows})
    fig, axes = plt.subplots(1, max(1, len(aggs)), figsize=(3.2 * max(1, len(aggs)), 3.0), sharex=True, sharey=True,
                             squeeze=False)
    for ax, agg in zip(axes[0], aggs):
        series = defaultdict(lambda: defaultdict(list))
        for r in rows:
            if r["Aggregator"] == agg:
                series[r["Momentum"]][r["num_malicious_clients"]].append(r["Accuracy"])
        for mom, pts in sorted(series.items()):
            xs = sorted(pts)
            mean = [sum(pts[x]) / len(pts[x]) for x in xs]
            ax.plot(xs, mean, marker="o", label=f"momentum {mom}")
            ax.fill_between(xs, [min(pts[x]) for x in xs], [max(pts[x]) for x in xs], alpha=0.2)
        ax.set_title(agg)
        ax.set_xlabel("num_malicious_clients")
    axes[0][0].set_ylabel("Accuracy")
    axes[0][-1].legend(fontsize=7)
    fig.tight_layout()
    if out:
        fig.savefig(out, dpi=150)
    if show:
        plt.show()
    return fig


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-dir", default="~/blades_results/fedavg_blades")
    ap.add_argument("--metric", default="acc_top_1")
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    rows = collect(a.log_dir, a.metric)
    if not rows:
        raise SystemExit(f"no finished trials under {a.log_dir}")
    plot(rows, out=a.out, show=a.out is None)


if __name__ == "__main__":
    main()
