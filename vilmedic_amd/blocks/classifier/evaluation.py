"""ref: vilmedic/blocks/classifier/evaluation.py:7-60 (logits / labels collector)."""
import numpy as np
import torch


def evaluation(models, config, dl, **kwargs):
    logits = labels = losses = None
    cumulative_index = 0
    for num_batch, batch in enumerate(dl):
        label = batch["labels"]
        batch_size = label.shape[0]
        batch = {k: v.cuda() if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        results = [model(**batch) for model in models]
        num_classes = results[0]["output"].shape[-1]
        if num_batch == 0:
            logits = np.zeros((len(dl.dataset), len(models), num_classes))
            labels = np.zeros((len(dl.dataset),) + tuple(label.shape[1:]))
            losses = np.zeros((len(dl), len(models)))
        for j, r in enumerate(results):
            logits[cumulative_index:cumulative_index + batch_size, j] = r["output"].float().data.cpu().numpy()
            losses[num_batch][j] = r["loss"].cpu().item()
        labels[cumulative_index:cumulative_index + batch_size] = label.data.cpu().numpy()
        cumulative_index += batch_size
    preds = np.mean(logits, axis=1)
    return {"loss": np.mean(losses), "refs": labels, "hyps": preds, "logits": logits}
