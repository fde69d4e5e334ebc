"""mmcv-free `Config.fromfile` for the reference's python-file configs (mmcv.Config is absent here):
executes the file, keeps its public names in an attribute dict."""
import os


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on construction."""

    def __init__(self, *args, **kwargs):
        super(ConfigDict, self).__init__(*args, **kwargs)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, ConfigDict):
                dict.__setitem__(self, k, ConfigDict(v))
            elif isinstance(v, (list, tuple)) and any(isinstance(x, dict) for x in v):
                dict.__setitem__(self, k, type(v)(ConfigDict(x) if isinstance(x, dict) else x for x in v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def copy(self):
        return ConfigDict(dict.copy(self))


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


class Config(object):

    def __init__(self, cfg_dict=None, filename=None, text=''):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)
        object.__setattr__(self, '_text', text)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        if not filename.endswith('.py'):
            raise IOError('Only py type is supported')
        text = open(filename, 'r').read()
        scope = {'__file__': filename}
        exec(compile(text, filename, 'exec'), scope)
        cfg = {k: v for k, v in scope.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        return Config(cfg, filename=filename, text=text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __repr__(self):
        return 'Config (path: {}): {}'.format(self._filename, dict.__repr__(self._cfg_dict))
