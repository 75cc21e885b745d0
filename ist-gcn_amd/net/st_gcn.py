"""Drop-in for the reference's `net/st_gcnold.py`: same `Model(in_channels, num_class, graph_args,
edge_importance_weighting, **kwargs)` / `st_gcn(...)` classes, state_dict keys and forward semantics, computed
by the MI355X HIP kernels (see net/_model.py for the variant table and file:line map)."""
from ._model import make

Model, st_gcn = make('st_gcn')
