def get_n_params(module):
    """ref: vilmedic/models/utils.py (parameter count string)."""
    pp = sum(p.numel() for p in module.parameters())
    return "Number of parameters: {:,}".format(pp)
