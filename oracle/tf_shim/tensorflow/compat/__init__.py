class _V1:
    def __getattr__(self, name):
        raise NotImplementedError(f"tf.compat.v1.{name} is not provided by the stand-in")


v1 = _V1()
