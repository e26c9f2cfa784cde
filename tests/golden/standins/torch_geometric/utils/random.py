def erdos_renyi_graph(*a, **k):
    raise NotImplementedError('random edges are disabled in the reference (random_probability = 0)')
