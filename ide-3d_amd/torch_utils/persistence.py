"""`persistent_class` decorator (light version of the reference's torch_utils/persistence.py:35).

The reference embeds module *source code* into pickles; the render path only needs the recorded
constructor arguments (`init_args` / `init_kwargs`, persistence.py:110-116) so that a module can be
re-created and `misc.copy_params_and_buffers` applied.  This keeps that contract and nothing else.
"""

import copy


def persistent_class(orig_class):
    class Decorator(orig_class):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)

        @property
        def init_args(self):
            return copy.deepcopy(self._init_args)

        @property
        def init_kwargs(self):
            return copy.deepcopy(self._init_kwargs)

    Decorator.__name__ = orig_class.__name__
    Decorator.__qualname__ = orig_class.__qualname__
    Decorator.__module__ = orig_class.__module__
    Decorator.__doc__ = orig_class.__doc__
    return Decorator
