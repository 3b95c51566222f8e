class List(list):
    """numba.typed.List stand-in: a plain list with ``empty_list``."""

    @classmethod
    def empty_list(cls, item_type=None):
        return cls()
