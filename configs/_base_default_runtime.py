# shared runtime settings (the role of the reference's configs/_base_/default_runtime.py)
log_config = dict(interval=5)
checkpoint_config = dict(interval=1)
workflow = [('train', 1)]
dist_params = dict(backend='nccl')
