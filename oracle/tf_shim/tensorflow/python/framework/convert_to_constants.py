def convert_variables_to_constants_v2_as_graph(*a, **k):
    raise NotImplementedError("graph freezing is not provided by the stand-in")
