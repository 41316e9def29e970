"""Sequential restatement of the reference's training generator (TEST INFRASTRUCTURE, see oracle/__init__).

Follows Proposed_Work_Results.py:49-270 step by step for the 3-class MTL configuration -- one file at a time, buffers grown
with np.append, the first batchSize rows of every class buffer handed out, labels built row by row -- with the two
per-file calls (`get_featuregram`, `get_feature_patches`) as parameters, so that the product's generator (which decides
the files of a batch first and runs them together) can be compared batch by batch under the same numpy random state."""
from __future__ import annotations

import os

import numpy as np


def reference_generator(PARAMS, folder, file_list, batchSize, featuregram_fn, patches_fn):
    np.random.shuffle(file_list['speech'])                       # :51
    np.random.shuffle(file_list['music'])                        # :52
    todo = {'speech': list(file_list['speech']), 'music': list(file_list['music'])}
    held = {'speech': None, 'music': None, 'mix': None}
    count = {'speech': 0, 'music': 0, 'mix': 0}
    np.random.shuffle(file_list['speech+music'])                 # :67
    todo['mix'] = list(file_list['speech+music'])
    mix_db = np.empty([], dtype=float)
    model = PARAMS['Model']
    n_fft, n_mels, featName = PARAMS['n_fft'][model], PARAMS['n_mels'][model], PARAMS['featName'][model]

    def grow(key, patches):
        held[key] = patches if count[key] == 0 else np.append(held[key], patches, axis=0)   # :113-116
        count[key] += np.shape(patches)[0]

    while True:
        for key, cls in (('speech', 'speech'), ('music', 'music')):                       # :84-156
            while count[key] < batchSize:
                if not todo[key]:
                    todo[key] = list(file_list[key])
                name = todo[key].pop()
                path = folder + '/' + cls + '/' + name
                if not os.path.exists(path):
                    continue
                args = (path, '') if key == 'speech' else ('', path)
                fv = featuregram_fn(PARAMS, cls, PARAMS['feature_opDir'], args[0], args[1], None, n_fft, n_mels, featName)
                grow(key, patches_fn(PARAMS, fv, PARAMS['W'], PARAMS['W_shift'], featName))
        batch = held['music'][:batchSize, :]                                              # :161-162
        batch = np.append(batch, held['speech'][:batchSize, :], axis=0)
        for key in ('music', 'speech'):                                                   # :164-168
            count[key] -= batchSize
            held[key] = held[key][batchSize:, :]
        cls_id = [0] * batchSize + [1] * batchSize                                        # :170-171
        smr = np.ones((3 * batchSize, 2))
        smr[:batchSize] = [1, 0]
        smr[batchSize:2 * batchSize] = [0, 1]
        while count['mix'] < batchSize:                                                   # :177-218
            if not todo['mix']:
                todo['mix'] = list(file_list['speech+music'])
            np.random.shuffle(todo['mix'])                                                # :180
            info = todo['mix'].pop()
            sp, mu = folder + '/speech/' + info['speech'], folder + '/music/' + info['music']
            if not (os.path.exists(sp) and os.path.exists(mu)):
                continue
            fv = featuregram_fn(PARAMS, 'speech_music', PARAMS['feature_opDir'], sp, mu, info['SMR'], n_fft, n_mels, featName)
            p = patches_fn(PARAMS, fv, PARAMS['W'], PARAMS['W_shift'], featName)
            db = np.array([info['SMR']] * np.shape(p)[0])
            mix_db = db if count['mix'] == 0 else np.append(mix_db, db)
            grow('mix', p)
        batch = np.append(batch, held['mix'][:batchSize, :], axis=0)                      # :222
        count['mix'] -= batchSize
        held['mix'] = held['mix'][batchSize:, :]
        cls_id += [2] * batchSize
        for i in range(batchSize):                                                        # :227-231
            d = mix_db[i]
            smr[2 * batchSize + i] = [1 / np.power(10, d / 10), 1] if d >= 0 else [1, np.power(10, d / 10)]
        mix_db = mix_db[batchSize:]
        if 'Lemaire_et_al' in model:
            batch = np.transpose(batch, axes=(0, 2, 1))                                   # :235-236
        if PARAMS['data_augmentation_with_noise']:                                        # :239-242
            scale = np.random.choice([5e-3, 1e-3, 5e-4, 1e-4])
            batch = np.add(batch, np.random.normal(loc=0.0, scale=scale, size=np.shape(batch)))
        onehot = np.eye(len(PARAMS['classes']), dtype=np.float32)[cls_id]                 # :244
        S = np.array(cls_id)                                                              # :249-252
        S[:batchSize], S[batchSize:2 * batchSize], S[2 * batchSize:] = 0, 1, 0
        M = np.array(cls_id)                                                              # :257-260
        M[:batchSize], M[batchSize:2 * batchSize], M[2 * batchSize:] = 1, 0, 0
        yield batch, {'R': smr, 'S': S, 'M': M, '3C': onehot}                             # :262-268
